"""ctypes binding of libsegk.so (the C ABI declared in include/segk.h).

The product path has NO CPU or eager-PyTorch fallback: if the HIP library is missing this module raises
at first use, and every op raises RuntimeError(segk_last_error()) on a non-zero return code.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libsegk.so")

F32, BF16 = 0, 1
ABI_VERSION = 310          # SEGK_ABI_VERSION of the include/segk.h this table was written against
MAX_CLASSES = 8

_vp, _fp, _i, _l, _f, _d = C.c_void_p, C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_double

# name -> (restype, argtypes); mirrors include/segk.h one to one (checked by tests/test_abi.py)
SIGNATURES = {
    "segk_version": (_i, []),
    "segk_entry_count": (_i, []),
    "segk_clock_probe": (_i, [_vp, _i, _i, _i, _vp]),
    "segk_debug_poison_tickets": (_i, [C.c_uint64, _vp]),
    "segk_build_id": (C.c_char_p, []),
    "segk_last_error": (C.c_char_p, []),
    "segk_nchw_to_nhwc": (_i, [_fp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "segk_nhwc_to_nchw": (_i, [_vp, _fp, _i, _i, _i, _i, _i, _i, _vp]),
    "segk_pack_conv_weight": (_i, [_fp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "segk_pack_conv3x3_both": (_i, [_fp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "segk_pack_multi": (_i, [_vp, _i, _i, _i, _vp]),
    "segk_pack_convt_chunk": (_i, []),
    "segk_pack_convt_weight": (_i, [_fp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "segk_conv_tiles": (_i, [_i, _i, _i, _i, _i, _i]),
    "segk_bn_stats_floats": (_i, [_i, _i]),
    "segk_conv3x3": (_i, [_vp, _vp, _vp, _fp, _fp, _fp, _vp, _vp, _fp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "segk_conv_writes_act_q": (_i, [_i, _i, _i]),
    "segk_conv3x3_act": (_i, [_vp, _vp, _fp, _fp, _vp, _vp, _fp, _i, _i, _i, _i, _i, _i, _vp]),
    "segk_stem3x3_rows": (_i, [_i, _i, _i, _i, _i, _i]),
    "segk_stem3x3": (_i, [_fp, _fp, _vp, _vp, _fp, _i, _i, _i, _i, _i, _i, _vp]),
    "segk_stem3x3_wgrad_slabs": (_i, [_i, _i, _i, _i, _i, _i]),
    "segk_stem3x3_wgrad": (_i, [_fp, _vp, _fp, _i, _i, _i, _i, _i, _i, _vp]),
    "segk_conv1x1": (_i, [_vp, _vp, _fp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "segk_convt2x2_fwd": (_i, [_vp, _vp, _fp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "segk_convt2x2_dgrad": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "segk_wgrad_tiles": (_i, [_i, _i, _i, _i, _i]),
    "segk_wgrad_split": (_i, [_i, _i, _i, _i, _i, _i]),
    "segk_wgrad": (_i, [_vp, _vp, _vp, _fp, _fp, _fp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "segk_wgrad_reduce": (_i, [_fp, _i, _fp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "segk_wgrad_reduce_multi": (_i, [_vp, _i, _vp]),
    "segk_bn_finalize": (_i, [_fp, _i, _i, _i, _d, _fp, _fp, _fp, _fp, _fp, _f, _f, _i, _fp, _fp, _fp, _fp, _vp]),
    "segk_bn_relu_apply": (_i, [_vp, _vp, _fp, _fp, _l, _i, _i, _vp]),
    "segk_bn_relu_apply_pool": (_i, [_vp, _vp, _vp, _fp, _fp, _i, _i, _i, _i, _i, _vp]),
    "segk_bn_bwd_blocks": (_i, [_l, _i, _i]),
    "segk_bn_relu_bwd": (_i, [_vp, _vp, _vp, _fp, _fp, _fp, _fp, _l, _i, _i, _fp, _fp, _fp, _fp, _i, _vp]),
    "segk_channel_sum": (_i, [_vp, _l, _i, _i, _fp, _fp, _i, _vp]),
    "segk_maxpool2x2_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "segk_maxpool2x2_bwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "segk_maxpool_bwd_stat_blocks": (_i, [_i, _i, _i, _i, _i]),
    "segk_maxpool2x2_bwd_bnstat": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _fp, _fp, _fp, _fp, _fp, _vp, _i, _vp]),
    "segk_bn_relu_bwd_from_part": (_i, [_vp, _vp, _vp, _fp, _fp, _fp, _fp, _l, _i, _i, _fp, _i, _fp, _fp, _fp, _i, _vp]),
    "segk_bilinear_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "segk_bilinear_bwd": (_i, [_vp, _vp, _fp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "segk_linear": (_i, [_vp, _vp, _fp, _vp, _l, _i, _i, _i, _i, _vp]),
    "segk_vit_patchify": (_i, [_fp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "segk_vit_embed_ln": (_i, [_vp, _fp, _fp, _fp, _fp, _f, _fp, _i, _i, _i, _i, _i, _vp]),
    "segk_add_layernorm": (_i, [_fp, _vp, _fp, _fp, _f, _vp, _l, _i, _i, _i, _vp]),
    "segk_linear_splitk": (_i, [_vp, _vp, _fp, _vp, _l, _i, _i, _i, _i, _vp]),
    "segk_add_layernorm_parts": (_i, [_fp, _vp, _i, _l, _fp, _fp, _f, _vp, _l, _i, _i, _i, _vp]),
    "segk_attention": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _f, _i, _vp]),
    "segk_vit_tokens_to_grid": (_i, [_fp, _vp, _i, _i, _i, _i, _i, _vp]),
    "segk_resize_pad": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "segk_crop_resize": (_i, [_fp, _fp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "segk_head_fwd": (_i, [_vp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "segk_head_fwd_bn": (_i, [_vp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "segk_head_bwd_bn": (_i, [_fp, _vp, _fp, _vp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _fp, _fp, _fp, _fp, _fp, _i, _vp]),
    "segk_head_part_floats": (_i, [_l, _i]),
    "segk_head_bwd": (_i, [_fp, _vp, _fp, _vp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "segk_head_bwd_blocks": (_i, [_l]),
    "segk_head_bwd_bnstat": (_i, [_fp, _vp, _fp, _vp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _fp, _fp, _fp, _fp, _fp, _i, _vp]),
    "segk_loss_part_floats": (_i, [_l]),
    "segk_loss_state_floats": (_i, []),
    "segk_loss_fwd": (_i, [_fp, _vp, _fp, _i, _i, _l, _i, _f, _f, _f, _fp, _fp, _fp, _vp]),
    "segk_loss_bwd": (_i, [_fp, _vp, _fp, _fp, _fp, _i, _i, _l, _i, _f, _f, _fp, _vp]),
    "segk_prompt_mix_fwd": (_i, [_fp, _fp, _fp, _i, _l, _vp]),
    "segk_prompt_mix_bwd": (_i, [_fp, _fp, _fp, _fp, _i, _l, _vp]),
    "segk_prob_loss_fwd": (_i, [_fp, _vp, _fp, _i, _i, _l, _i, _f, _f, _f, _i, _f, _fp, _fp, _fp, _vp]),
    "segk_prob_loss_bwd": (_i, [_fp, _vp, _fp, _fp, _fp, _i, _i, _l, _i, _f, _f, _i, _f, _fp, _vp]),
    "segk_confusion": (_i, [_fp, _vp, _i, _i, _l, _vp, _vp]),
}



class ReduceJob(C.Structure):
    """segk_reduce_job of include/segk.h (one reduction of segk_wgrad_reduce_multi)."""
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("kind", C.c_int), ("S", C.c_int), ("N", C.c_int), ("CA", C.c_int),
                ("CB", C.c_int), ("Np", C.c_int), ("CAp", C.c_int), ("CBp", C.c_int), ("taps", C.c_int), ("pad_", C.c_int)]


_lib = None


def load():
    """Load libsegk.so (building nothing: run `python -m image_segmentation_amd.build` or
    __graft_entry__.build() first).  Raises if the library is absent -- there is no fallback path."""
    global _lib, LIB_PATH
    if _lib is None:
        if os.environ.get("SEGK_LIB"):       # diagnostic builds only (tools/stamp_build.sh, same-box A/B of two builds)
            LIB_PATH = os.path.abspath(os.environ["SEGK_LIB"])
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the HIP extension is required (no CPU/eager fallback exists). "
                "Build it with `python -m image_segmentation_amd.build`.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(lib, name)
            except AttributeError:
                raise RuntimeError(f"{LIB_PATH} does not export {name}: it was built from another include/segk.h than "
                                   "this binding (rebuild with `python -m image_segmentation_amd.build`)") from None
            fn.restype = res
            fn.argtypes = args
        # A library built against another header is refused (SEGK_LIB / --lib diagnostic builds included): the host sizes
        # buffers with this table's queries, so a kernel set with other geometry rules must never run behind it
        ver, nent = lib.segk_version(), lib.segk_entry_count()
        if ver != ABI_VERSION or nent != len(SIGNATURES):
            raise RuntimeError(f"{LIB_PATH}: ABI version {ver} with {nent} entries, this binding expects version "
                               f"{ABI_VERSION} with {len(SIGNATURES)} entries -- refusing to drive a library built from "
                               "another include/segk.h")
        _lib = lib
    return _lib


def call(name, *args):
    """Invoke an int-returning entry point; raise with the library's message on failure."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise RuntimeError(f"{name} failed ({rc}): {lib.segk_last_error().decode()}")


def build_id():
    """Source hash compiled into the loaded library (see build.source_hash)."""
    return load().segk_build_id().decode()


def query(name, *args):
    """Invoke a pure size/geometry query (returns its int result)."""
    return getattr(load(), name)(*args)
