"""Host side of the fused segmentation ops: torch.autograd.Functions that drive the HIP kernels of
libsegk.so through the C ABI (include/segk.h).  PyTorch is used for device memory (caching allocator),
streams and autograd bookkeeping only -- every FLOP and byte of the hot path runs in the hand-written
kernels.  There is no CPU / eager fallback: non-CUDA tensors raise.

Activation convention ("act tensor"): logical shape [B,C,H,W] (what the reference's nn.Modules exchange)
backed by an NHWC buffer [B,H,W,Cp] with Cp = C rounded up to 32 and zero padding channels, in the compute
dtype (fp32 = parity mode, bf16 = performance mode).  For C % 32 == 0 this is exactly torch's
channels_last memory format.
"""
import os
import sys
import weakref

import torch

from . import _lib

_DT = {torch.float32: _lib.F32, torch.bfloat16: _lib.BF16}
_compute_dtype = torch.bfloat16


def set_compute_dtype(dtype):
    """torch.float32: exact-fp32 MFMA kernels (parity gate: logits within 1e-3 of the CPU reference);
    torch.bfloat16: bf16 storage + bf16 MFMA with fp32 accumulation and fp32 BatchNorm statistics."""
    global _compute_dtype
    if dtype not in _DT:
        raise ValueError("compute dtype must be torch.float32 or torch.bfloat16")
    _compute_dtype = dtype


def get_compute_dtype():
    return _compute_dtype


def pad32(c):
    return (c + 31) // 32 * 32


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """Handle of the current HIP stream of the current device.  The raw C entry points cost ~0.3 us; the
    torch.cuda.current_stream() object path costs ~4 us, 270 times per training step."""
    if _raw_stream is not None and _raw_device is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


def _require_cuda(t, what):
    if not t.is_cuda:
        raise RuntimeError(f"{what}: expected a CUDA/HIP tensor -- image_segmentation_amd has no CPU path "
                           "(the CPU oracle lives in oracle/ and is test infrastructure only)")


def _p(t):
    return 0 if t is None else t.data_ptr()


# ------------------------------------------------------------------------------------------------
# act-tensor helpers
def new_act(B, C, H, W, dtype, device):
    """Allocate an act tensor; padding channels are zeroed only when they exist."""
    Cp = pad32(C)
    buf = torch.empty((B, H, W, Cp), dtype=dtype, device=device)
    if Cp != C:
        buf[..., C:].zero_()
    return buf.permute(0, 3, 1, 2)[:, :C]


def act_view(buf, C):
    return buf.permute(0, 3, 1, 2)[:, :C]


def act_info(t, dtype):
    """Return (data_ptr, Cp) if `t` already is an act tensor of `dtype`, else None."""
    if t.dim() != 4 or t.dtype != dtype or not t.is_cuda:
        return None
    B, C, H, W = t.shape
    sb, sc, sh, sw = t.stride()
    Cp = sw if W > 1 else (sh if H > 1 else pad32(C))
    if sc != 1 or Cp < C or Cp % 32 != 0 or Cp != pad32(C):
        return None
    if (W > 1 and sw != Cp) or (H > 1 and sh != W * Cp) or (B > 1 and sb != H * W * Cp):
        return None
    if (t.storage_offset() * t.element_size()) % 16 != 0:
        return None
    return t.data_ptr(), Cp


def to_act(x, dtype):
    """Convert any [B,C,H,W] CUDA tensor to an act tensor (zero-copy when it already is one)."""
    _require_cuda(x, "to_act")
    if act_info(x, dtype) is not None:
        return x
    B, C, H, W = x.shape
    src = x.detach()
    if C % 32 == 0 and src.stride(1) == 1 and src.permute(0, 2, 3, 1).is_contiguous():
        # already dense NHWC (e.g. ViT token grids), only the dtype differs: cast in place of a re-layout
        return act_view(src.permute(0, 2, 3, 1).to(dtype), C)
    if src.dtype != torch.float32 or not src.is_contiguous():
        src = src.float().contiguous()       # layout/dtype normalisation of a foreign tensor (edge only)
    out = torch.empty((B, H, W, pad32(C)), dtype=dtype, device=x.device)
    _lib.call("segk_nchw_to_nhwc", src.data_ptr(), out.data_ptr(), B, C, H, W, pad32(C), _DT[dtype], _stream())
    return act_view(out, C)


def act_to_nchw(t):
    """act tensor -> contiguous NCHW fp32 (what the reference modules would return)."""
    info = act_info(t, t.dtype)
    if info is None:
        return t.float().contiguous()
    B, C, H, W = t.shape
    out = torch.empty((B, C, H, W), dtype=torch.float32, device=t.device)
    _lib.call("segk_nhwc_to_nchw", info[0], out.data_ptr(), B, C, H, W, info[1], _DT[t.dtype], _stream())
    return out


def _private_act_copy(t, dtype):
    """A fresh NHWC buffer [B,H,W,Cp] holding the bytes of act tensor `t` (padding channels included)."""
    ptr, Cp = act_info(t, dtype)
    B, C, H, W = t.shape
    n = B * H * W * Cp
    out = torch.empty((B, H, W, Cp), dtype=dtype, device=t.device)
    out.view(-1).copy_(torch.as_strided(t, (n,), (1,), t.storage_offset()))
    return out


def _raw(t, dtype):
    """(ptr, Cp) of an act tensor, converting if needed; returns (tensor_kept_alive, ptr, Cp)."""
    t = to_act(t, dtype)
    ptr, Cp = act_info(t, dtype)
    return t, ptr, Cp


# ------------------------------------------------------------------------------------------------
# packed-weight cache (parameters stay fp32 OIHW/IOHW: the source of truth for state_dict/optimizer)
# Optimizer steps invalidate every packed copy of a trainable parameter.  The autograd version counter alone is not
# enough: fused optimizers (torch.optim.AdamW(fused=True), torch._fused_adamw_) update parameters in place WITHOUT
# moving `param._version`, so a global post-step hook counts optimizer steps as well.
INPLACE_SKIP_GRAD = True   # the pooling backward accumulates into the incoming skip gradient's storage (False: private copy)
FUSE_BN_REDUCE = True    # pooling / head backward also accumulate the producing block's BN2 backward reductions
# a block followed by the output head hands it the pre-activation: BN+ReLU happen inside the head kernels
HEAD_ON_Z = os.environ.get("SEGK_HEAD_ON_Z", "1") != "0"       # the env switch is for same-box A/B runs of bench.py
_OPT_EPOCH = [0]


def _on_optimizer_step(optimizer, args, kwargs):
    _OPT_EPOCH[0] += 1


from torch.optim.optimizer import register_optimizer_step_post_hook as _register_step_hook  # noqa: E402

_register_step_hook(_on_optimizer_step)


def invalidate_packed_weights():
    """Call after changing parameters behind autograd's back (e.g. writing through `.data`)."""
    _OPT_EPOCH[0] += 1


class PackCache:
    """Re-packs a parameter into the MFMA layout when its autograd version counter moved, when it was re-allocated,
    or (trainable parameters) when any optimizer has stepped since the copy was made."""

    def __init__(self):
        self._c = {}
        self.multi = {}      # key_a -> (kind, key_b, param, d0, d1, dtype): what repack_stale() refreshes in one launch

    def get_pair(self, key_a, key_b, param, builder, meta=None):
        """Two packed copies made by one builder call (returns the first; the second is served by get(key_b)).
        meta=(kind, d0, d1, dtype) registers the pair for repack_stale() (kind 0: 3x3 weight, d = CA, CB; kind 1:
        ConvTranspose weight)."""
        ver = (param._version, param.data_ptr(), param.device, _OPT_EPOCH[0] if param.requires_grad else -1)
        hit = self._c.get(key_a)
        if hit is None or hit[0] != ver:
            a, b = builder()
            hit = (ver, a)
            self._c[key_a] = hit
            self._c[key_b] = (ver, b)
            if meta is not None:
                self.multi[key_a] = (meta[0], key_b, param) + tuple(meta[1:])
        return hit[1]

    def get_registered(self, key, param, builder, meta):
        """get() that also registers the single copy for repack_stale() (kind 2: ConvTranspose bias operand)."""
        self.multi[key] = (meta[0], None, param) + tuple(meta[1:])
        return self.get(key, param, builder)

    def get(self, key, param, builder):
        ver = (param._version, param.data_ptr(), param.device, _OPT_EPOCH[0] if param.requires_grad else -1)
        hit = self._c.get(key)
        if hit is None or hit[0] != ver:
            hit = (ver, builder())
            self._c[key] = hit
        return hit[1]


def pack_conv(w, CA, CB, dtype, mode, taps=9):
    Cout = w.shape[0]
    Coutp, CAp, CBp = pad32(Cout), pad32(CA), (pad32(CB) if CB else 0)
    dst = torch.empty(((CAp + CBp) * taps * Coutp,), dtype=dtype, device=w.device)
    wf = w.detach()
    if wf.dtype != torch.float32 or not wf.is_contiguous():
        wf = wf.float().contiguous()
    _lib.call("segk_pack_conv_weight", wf.data_ptr(), dst.data_ptr(), Cout, CA, CB, Coutp, CAp, CBp, taps, mode,
              _DT[dtype], _stream())
    return dst


def pack_conv_both(w, CA, CB, dtype):
    """Forward and data-gradient layouts of a 3x3 weight in one kernel pass -> (fwd, dgrad)."""
    Cout = w.shape[0]
    Coutp, CAp, CBp = pad32(Cout), pad32(CA), (pad32(CB) if CB else 0)
    n = (CAp + CBp) * 9 * Coutp
    both = torch.empty((2 * n,), dtype=dtype, device=w.device)
    wf = w.detach()
    if wf.dtype != torch.float32 or not wf.is_contiguous():
        wf = wf.float().contiguous()
    _lib.call("segk_pack_conv3x3_both", wf.data_ptr(), both.data_ptr(), both[n:].data_ptr(), Cout, CA, CB, Coutp, CAp, CBp,
              _DT[dtype], _stream())
    return both[:n], both[n:]


_PACK_CONVT_CHUNK = [0]


def _entry_geometry(kind, param, d0, d1):
    """(elements of each destination, words 3..6 of the table entry without block0, blocks) of one registered tensor."""
    if kind == 0:            # Conv2d 3x3 [Cout][CA+CB][3][3]: d0, d1 = CA, CB
        Cout = param.shape[0]
        Coutp, CAp, CBp = pad32(Cout), pad32(d0), (pad32(d1) if d1 else 0)
        return (CAp + CBp) * 9 * Coutp, (Cout, d0, d1, Coutp, CAp, CBp), (CAp + CBp) // 32 * (Coutp // 32)
    if kind == 1:            # ConvTranspose2d [Cin][Cout][2][2]
        Cin, Cout = param.shape[0], param.shape[1]
        Cinp, Coutp = pad32(Cin), pad32(Cout)
        if not _PACK_CONVT_CHUNK[0]:
            _PACK_CONVT_CHUNK[0] = _lib.query("segk_pack_convt_chunk")
        n = Cinp * 4 * Coutp
        return n, (Cout, Cin, 0, Coutp, Cinp, 0), (n + _PACK_CONVT_CHUNK[0] - 1) // _PACK_CONVT_CHUNK[0]
    if kind == 3:            # Conv2d 1x1 weight [Cout][Cin][1][1]: d0 = Cin (forward layout only: key_b is None)
        Cout = param.shape[0]
        Coutp, CAp = pad32(Cout), pad32(d0)
        if not _PACK_CONVT_CHUNK[0]:
            _PACK_CONVT_CHUNK[0] = _lib.query("segk_pack_convt_chunk")
        n = CAp * Coutp
        return n, (Cout, d0, 0, Coutp, CAp, 0), (n + _PACK_CONVT_CHUNK[0] - 1) // _PACK_CONVT_CHUNK[0]
    Cout = param.shape[0]    # kind 2: bias -> fp32 [reps][Coutp]; d0 = reps (0: the 4 of a ConvTranspose2d bias, 1: a conv bias)
    reps = d0 if d0 > 0 else 4
    return reps * pad32(Cout), (Cout, d0, 0, pad32(Cout), 0, 0), 1


def _pack_caches(model):
    """Every PackCache under `model`: registered sub-modules and the fused drivers a module keeps OUT of the module tree
    (object.__setattr__: the CLIP decoder's `_dc` / `_skip` / `_init`, which share their parameters with registered
    children) -- without the second kind the decoder's weights were re-packed by one launch each."""
    seen, out, stack = set(), [], list(model.modules())
    while stack:
        m = stack.pop()
        if id(m) in seen:
            continue
        seen.add(id(m))
        c = getattr(m, "cache", None)
        if isinstance(c, PackCache):
            out.append(c)
        for v in vars(m).values():
            if isinstance(v, torch.nn.Module) and id(v) not in seen:
                stack.extend(v.modules())
    return out


def repack_stale(model):
    """Refresh every stale packed copy registered under `model` with ONE launch (segk_pack_multi) instead of one per tensor:
    the (forward, data-gradient) pairs of the 3x3 and ConvTranspose weights and the ConvTranspose bias operands.  Called at
    the top of a model's forward, it does nothing until an optimizer step (or invalidate_packed_weights) moved the epoch.
    The tensors are those the autograd nodes registered on their first training forward (PackCache.multi); their
    destination buffers are re-used (every consumer is ordered on the launch stream)."""
    st = model.__dict__.get("_segk_pack_state")
    if st is None:
        st = model.__dict__["_segk_pack_state"] = {"epoch": _OPT_EPOCH[0], "caches": None, "tables": {}}
        return                      # nothing can be registered before the first forward
    if st["epoch"] == _OPT_EPOCH[0]:
        return
    st["epoch"] = _OPT_EPOCH[0]
    if st["caches"] is None:
        st["caches"] = _pack_caches(model)
    groups = {}
    for cache in st["caches"]:
        for key_a, (kind, key_b, param, d0, d1, dtype) in cache.multi.items():
            ha = cache._c.get(key_a)
            hb = cache._c.get(key_b) if key_b is not None else None
            if ha is None or (key_b is not None and hb is None):
                continue
            if not param.is_cuda or param.dtype != torch.float32 or not param.is_contiguous():
                continue
            ver = (param._version, param.data_ptr(), param.device, _OPT_EPOCH[0] if param.requires_grad else -1)
            if ha[0] == ver and (hb is None or hb[0] == ver):
                continue
            n, _, _ = _entry_geometry(kind, param, d0, d1)
            want = torch.float32 if kind == 2 else dtype
            if ha[1].numel() != n or ha[1].dtype != want or not ha[1].is_contiguous():
                continue
            if hb is not None and (hb[1].numel() != n or hb[1].dtype != want or not hb[1].is_contiguous()):
                continue
            groups.setdefault((dtype, param.device), []).append(
                (cache, key_a, key_b, ver, param, ha[1], None if hb is None else hb[1], kind, d0, d1))
    for (dtype, dev), items in groups.items():
        if len(items) < 2:
            continue                 # a single tensor goes through its own path
        for c0 in range(0, len(items), 64):
            chunk = items[c0:c0 + 64]
            sig = tuple((it[4].data_ptr(), it[5].data_ptr(), 0 if it[6] is None else it[6].data_ptr(), it[7], it[8], it[9])
                        for it in chunk)
            tab = st["tables"].get((dtype, dev, c0))
            if tab is None or tab[0] != sig:
                import numpy as np
                words = np.zeros((len(chunk), 8), dtype=np.int64)
                ints = words.view(np.int32).reshape(len(chunk), 16)
                blk = 0
                for i, (_, _, _, _, param, a, b, kind, d0, d1) in enumerate(chunk):
                    _, dims, nblk = _entry_geometry(kind, param, d0, d1)
                    words[i, 0], words[i, 1], words[i, 2] = param.data_ptr(), a.data_ptr(), 0 if b is None else b.data_ptr()
                    ints[i, 6:12] = dims
                    ints[i, 12], ints[i, 13] = blk, kind
                    blk += nblk
                tab = (sig, torch.from_numpy(words).to(dev), blk)
                st["tables"][(dtype, dev, c0)] = tab
            with torch.cuda.device(dev):
                _lib.call("segk_pack_multi", tab[1].data_ptr(), len(chunk), tab[2], _DT[dtype], _stream())
            for cache, key_a, key_b, ver, _, a, b, _, _, _ in chunk:
                cache._c[key_a] = (ver, a)
                if key_b is not None:
                    cache._c[key_b] = (ver, b)


def pack_convt(w, dtype, mode):
    Cin, Cout = w.shape[0], w.shape[1]
    dst = torch.empty((pad32(Cin) * 4 * pad32(Cout),), dtype=dtype, device=w.device)
    wf = w.detach()
    if wf.dtype != torch.float32 or not wf.is_contiguous():
        wf = wf.float().contiguous()
    _lib.call("segk_pack_convt_weight", wf.data_ptr(), dst.data_ptr(), Cin, Cout, pad32(Cin), pad32(Cout), mode,
              _DT[dtype], _stream())
    return dst


# ------------------------------------------------------------------------------------------------
# optional per-kernel-family timing (bench.py): HIP events recorded on the launch stream around each call
class KernelTimer:
    """Collects (start, end) HIP-event pairs plus algorithmic flops/bytes per kernel family.  Events are
    recorded on torch's current stream, which is the stream every segk launch uses."""

    def __init__(self):
        self.rows = {}
        self.levels = {}      # (level name, phase) -> the same record, for spans opened inside a named DoubleConv node

    def span(self, tag, flops=0.0, nbytes=0.0):
        return _Span(self, tag, flops, nbytes)

    @staticmethod
    def _sum(rows):
        out = {}
        for key, r in rows.items():
            ms = sum(a.elapsed_time(b) for a, b in r["ev"])
            out[key] = {"launches": len(r["ev"]), "ms": ms, "flops": r["flops"], "bytes": r["bytes"]}
        return out

    def summary(self):
        torch.cuda.synchronize()
        return self._sum(self.rows)

    def level_summary(self):
        """{(level, phase): record}; phase = fwd | dgrad | wgrad (the 3x3 conv kernels of the block) | other."""
        torch.cuda.synchronize()
        return self._sum(self.levels)


class _Span:
    def __init__(self, timer, tag, flops, nbytes):
        self.t, self.tag, self.flops, self.nbytes = timer, tag, flops, nbytes

    def __enter__(self):
        self.a = torch.cuda.Event(enable_timing=True)
        self.a.record()

    def __exit__(self, *exc):
        b = torch.cuda.Event(enable_timing=True)
        b.record()
        r = self.t.rows.setdefault(self.tag, {"ev": [], "flops": 0.0, "bytes": 0.0})
        r["ev"].append((self.a, b))
        r["flops"] += self.flops
        r["bytes"] += self.nbytes
        if _LEVEL is not None:
            phase = {"conv3x3_igemm": "dgrad" if _LEVEL[1] else "fwd", "wgrad3x3": "wgrad"}.get(self.tag, "other")
            r = self.t.levels.setdefault((_LEVEL[0], phase), {"ev": [], "flops": 0.0, "bytes": 0.0})
            r["ev"].append((self.a, b))
            r["flops"] += self.flops
            r["bytes"] += self.nbytes


class _NoSpan:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NOSPAN = _NoSpan()
TIMER = None          # set to a KernelTimer() to collect per-kernel timings
_LEVEL = None         # (level name, in backward) while a named DoubleConv node runs with a TIMER installed


def name_levels(model):
    """Give every DoubleConv block of `model` its module path as level name (bench.py: per-level roofline report)."""
    for name, m in model.named_modules():
        if hasattr(m, "bn_modules") and hasattr(m, "cache"):
            m.__dict__["_segk_level"] = name or "block"


class _level_scope:
    def __init__(self, mod, backward):
        self.v = (mod.__dict__.get("_segk_level"), backward) if TIMER is not None else None

    def __enter__(self):
        global _LEVEL
        self.prev = _LEVEL
        if self.v is not None and self.v[0] is not None:
            _LEVEL = self.v

    def __exit__(self, *exc):
        global _LEVEL
        _LEVEL = self.prev
        return False


def clock_probe(launches=4, blocks=256, iters=20000, device=None):
    """Median shader clock (GHz) the chip holds under a dense bf16 MFMA probe (segk_clock_probe: `launches` back-to-back
    launches of `iters` rounds per wave, the last one evaluated), for both MFMA shapes the kernels use.  Diagnostic only:
    numbers that explain box-to-box spread of the MFMA-bound kernels; it synchronises the device."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    res = {"method": f"s_memtime / s_memrealtime (100 MHz) around {iters} rounds of 16 v_mfma_f32_32x32x16_bf16 (or 32 "
                     f"v_mfma_f32_16x16x32_bf16: the same matrix work) per wave, one wave per SIMD on {blocks} workgroups, "
                     f"launch {launches} of {launches} back to back; pseudo-random register operands"}
    for shape, name in ((0, "mfma_32x32x16"), (1, "mfma_16x16x32")):
        out = torch.zeros((blocks * 4 * 2,), dtype=torch.int64, device=dev)
        with torch.cuda.device(dev):
            for _ in range(launches):
                _lib.call("segk_clock_probe", out.data_ptr(), blocks, iters, shape, _stream())
            torch.cuda.synchronize(dev)
        t = out.view(-1, 2).double().cpu()
        ghz = (t[:, 0] / t[:, 1].clamp(min=1.0)) * 0.1
        ms = float(t[:, 1].median()) / 1e5
        res[name] = {"median_ghz": round(float(ghz.median()), 4), "min_ghz": round(float(ghz.min()), 4),
                     "max_ghz": round(float(ghz.max()), 4), "probe_ms": round(ms, 3),
                     "probe_pflops": round(blocks * 4 * iters * 16 * 32768.0 / (ms * 1e-3) / 1e15, 3)}
    res["median_ghz"] = res["mfma_32x32x16"]["median_ghz"]
    return res


def _span(tag, flops=0.0, nbytes=0.0):
    return _NOSPAN if TIMER is None else TIMER.span(tag, flops, nbytes)


def _es(dtype):
    return 2 if dtype == torch.bfloat16 else 4


# ------------------------------------------------------------------------------------------------
# thin wrappers over the C ABI (all asynchronous on the current stream)
def _f32(n, dev):
    return torch.empty((n,), dtype=torch.float32, device=dev)


_ZERO_PAGES = {}


def _zero_page(dev):
    """64 zero bytes per device: the halo source of the LDS-DMA weight-gradient kernel."""
    key = torch.device(dev)
    if key.type == "cuda" and key.index is None:
        key = torch.device("cuda", torch.cuda.current_device())
    z = _ZERO_PAGES.get(key)
    if z is None:
        z = torch.zeros(256, dtype=torch.uint8, device=key)
        _ZERO_PAGES[key] = z
    return z


def conv3x3(srcA, ptrA, CAp, ptrB, CBp, wpacked, out_ptr, CO1p, out2_ptr, CO2p, B, H, W, dtype, scale=None,
            shift=None, stats=None, alg=None, tag="conv3x3_igemm"):
    """alg = (logical Cin, logical Cout) for the algorithmic flop/byte count (defaults to the padded sizes)."""
    cin, cout = alg if alg else (CAp + CBp, CO1p + CO2p)
    P, e = B * H * W, _es(dtype)
    with _span(tag, 2.0 * P * 9 * cin * cout, P * (cin + cout) * e + 9.0 * cin * cout * e):
        _lib.call("segk_conv3x3", ptrA, ptrB, wpacked.data_ptr(), 0, _p(scale), _p(shift), out_ptr, out2_ptr,
                  _p(stats), B, H, W, CAp, CBp, CO1p, CO2p, _DT[dtype], _stream())


def wgrad(dz_ptr, CDp, ptrA, CAp, ptrB, CBp, B, H, W, geo, dtype, dev, scale=None, shift=None, alg=None):
    """Returns the slabs tensor and S (split-K factor)."""
    taps = {0: 9, 1: 1, 2: 4}[geo]
    cd, ck = alg if alg else (CDp, CAp + CBp)
    P, e = B * H * W, _es(dtype)
    pk = P * (4 if geo == 2 else 1)
    tiles = _lib.query("segk_wgrad_tiles", B, H, W, geo, _DT[dtype])
    S = _lib.query("segk_wgrad_split", tiles, CDp, CAp, CBp, geo, _DT[dtype])
    slabs = _f32(S * CDp * taps * (CAp + CBp), dev)
    tag = {0: "wgrad3x3", 1: "wgrad1x1", 2: "wgrad_convt"}[geo]
    with _span(tag, 2.0 * P * taps * cd * ck, (P * cd + pk * ck) * e + 4.0 * taps * cd * ck):
        _lib.call("segk_wgrad", dz_ptr, ptrA, ptrB, _p(scale), _p(shift), slabs.data_ptr(), _zero_page(dev).data_ptr(),
                  S, B, H, W, CDp, CAp, CBp, geo, _DT[dtype], _stream())
    return slabs, S


_GRAD_DEST = None       # callback(param) -> fresh fp32 view to write the gradient into, or None (parallel.GradSync)
_GRAD_DEST_USED = set()


def grad_destination_begin(callback):
    """Between begin and end, weight gradients are written straight into the buffer `callback(param)` returns (the
    parameter's slice of its all-reduce bucket), each slice at most once (a weight used twice gets one slice and one
    private buffer; autograd sums them)."""
    global _GRAD_DEST
    _GRAD_DEST = callback
    _GRAD_DEST_USED.clear()


def grad_destination_end():
    global _GRAD_DEST
    _GRAD_DEST = None
    _GRAD_DEST_USED.clear()


def _grad_buffer(param, shape, dev):
    if _GRAD_DEST is not None and param is not None and param.data_ptr() not in _GRAD_DEST_USED:
        d = _GRAD_DEST(param)
        if d is not None and tuple(d.shape) == tuple(shape) and d.dtype == torch.float32:
            _GRAD_DEST_USED.add(param.data_ptr())
            return d
    return torch.empty(shape, dtype=torch.float32, device=dev)


def wgrad_to_param(slabs, S, shape, N, CA, CB, taps, dev, param=None):
    grad = _grad_buffer(param, shape, dev)
    with _span("wgrad_reduce", 0.0, 4.0 * (S + 1) * grad.numel()):
        _lib.call("segk_wgrad_reduce", slabs.data_ptr(), S, grad.data_ptr(), N, CA, CB, pad32(N), pad32(CA),
                  pad32(CB) if CB else 0, taps, _stream())
    return grad


class ReduceBatch:
    """Fixed-order reductions collected and run by ONE launch per four jobs (segk_wgrad_reduce_multi): the ConvTranspose
    weight-gradient slabs of an Up block together with that layer's bias gradient (column sums of the concat
    data-gradient's per-tile channel sums: no stock torch.sum).  The buffers handed out by the add_* methods hold their
    values once flush() has been called.  Only reductions whose inputs were produced just before belong in one batch: a
    slab buffer that waits for a later launch drops out of the Infinity Cache (measured: slower than separate launches)."""

    def __init__(self, dev):
        self.dev, self.jobs, self.keep, self.nbytes = dev, [], [], 0.0

    def add_weight(self, slabs, S, shape, N, CA, CB, taps, param=None):
        grad = _grad_buffer(param, shape, self.dev)
        self.jobs.append(_lib.ReduceJob(slabs.data_ptr(), grad.data_ptr(), 0, S, N, CA, CB, pad32(N), pad32(CA),
                                        pad32(CB) if CB else 0, taps, 0))
        self.keep.append(slabs)
        self.nbytes += 4.0 * (S + 1) * grad.numel()
        return grad

    def add_colsum(self, part, rows, ntot, col0, ncols):
        out = _f32(ncols, self.dev)
        self.jobs.append(_lib.ReduceJob(part.data_ptr(), out.data_ptr(), 1, rows, ntot, col0, ncols, 0, 0, 0, 0, 0))
        self.keep.append(part)
        self.nbytes += 8.0 * rows * ncols
        return out

    def flush(self):
        for i in range(0, len(self.jobs), 4):
            chunk = self.jobs[i:i + 4]
            arr = (_lib.ReduceJob * len(chunk))(*chunk)
            with _span("wgrad_reduce", 0.0, self.nbytes * len(chunk) / len(self.jobs)):
                _lib.call("segk_wgrad_reduce_multi", arr, len(chunk), _stream())
        self.jobs, self.keep, self.nbytes = [], [], 0.0


def bn_finalize(stats, tiles, C, count, conv_bias, bn_w, bn_b, rmean, rvar, momentum, eps, training, dev):
    Cp = pad32(C)
    scale, shift = _f32(Cp, dev), _f32(Cp, dev)
    mean, rstd = _f32(Cp, dev), _f32(Cp, dev)
    with _span("bn_finalize", 0.0, 8.0 * tiles * Cp if training else 0.0):
        _lib.call("segk_bn_finalize", _p(stats), tiles, Cp, C, float(count), _p(conv_bias), bn_w.data_ptr(),
                  bn_b.data_ptr(), _p(rmean), _p(rvar), float(momentum), float(eps), int(training), scale.data_ptr(),
                  shift.data_ptr(), mean.data_ptr(), rstd.data_ptr(), _stream())
    return scale, shift, mean, rstd


def bn_relu_bwd(dy_ptr, z_ptr, dz_ptr, scale, shift, mean, rstd, P, C, dtype, dev, ready=None, frozen=False):
    """ready = (partials, rows) when the kernel that wrote dy already accumulated the reductions
    (segk_maxpool2x2_bwd_bnstat): only finalize + apply run.
    frozen: the BatchNorm ran on its running statistics (eval mode), i.e. it is a fixed affine map: dz = scale * g with no
    mean terms, dgamma = sum(g * xhat), dbeta = sum(g).  Served by the same kernels in two calls (the reductions with
    their apply pass discarded, then an apply pass from all-zero partials) -- not a hot path."""
    Cp = pad32(C)
    dgamma, dbeta = _f32(C, dev), _f32(C, dev)
    if frozen:
        nb = _lib.query("segk_bn_bwd_blocks", P, Cp, _DT[dtype])
        part, coef = _f32(nb * Cp * 2, dev), _f32(2 * Cp, dev)
        scratch = torch.empty((P * Cp,), dtype=dtype, device=dev)
        _lib.call("segk_bn_relu_bwd", dy_ptr, z_ptr, scratch.data_ptr(), scale.data_ptr(), shift.data_ptr(), mean.data_ptr(),
                  rstd.data_ptr(), P, Cp, C, part.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), coef.data_ptr(),
                  _DT[dtype], _stream())
        zero_part = torch.zeros((Cp * 2,), dtype=torch.float32, device=dev)
        unused_g, unused_b = _f32(C, dev), _f32(C, dev)
        _lib.call("segk_bn_relu_bwd_from_part", dy_ptr, z_ptr, dz_ptr, scale.data_ptr(), shift.data_ptr(),
                  mean.data_ptr(), rstd.data_ptr(), P, Cp, C, zero_part.data_ptr(), 1, unused_g.data_ptr(), unused_b.data_ptr(),
                  coef.data_ptr(), _DT[dtype], _stream())
        return dgamma, dbeta
    if ready is not None:
        part, nb = ready
        coef = _f32(2 * Cp, dev)
        with _span("bn_relu_bwd", 0.0, 3.0 * P * C * _es(dtype)):     # apply only: 2 reads + 1 write
            _lib.call("segk_bn_relu_bwd_from_part", dy_ptr, z_ptr, dz_ptr, scale.data_ptr(), shift.data_ptr(),
                      mean.data_ptr(), rstd.data_ptr(), P, Cp, C, part.data_ptr(), nb, dgamma.data_ptr(), dbeta.data_ptr(),
                      coef.data_ptr(), _DT[dtype], _stream())
        return dgamma, dbeta
    nb = _lib.query("segk_bn_bwd_blocks", P, Cp, _DT[dtype])
    part, coef = _f32(nb * Cp * 2, dev), _f32(2 * Cp, dev)
    with _span("bn_relu_bwd", 0.0, 5.0 * P * C * _es(dtype)):     # reduce: 2 reads; apply: 2 reads + 1 write
        _lib.call("segk_bn_relu_bwd", dy_ptr, z_ptr, dz_ptr, scale.data_ptr(), shift.data_ptr(), mean.data_ptr(),
                  rstd.data_ptr(), P, Cp, C, part.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), coef.data_ptr(),
                  _DT[dtype], _stream())
    return dgamma, dbeta


def channel_sum(ptr, P, C, dtype, dev):
    Cp = pad32(C)
    nb = _lib.query("segk_bn_bwd_blocks", P, Cp, _DT[dtype])
    part, out = _f32(nb * Cp, dev), _f32(C, dev)
    _lib.call("segk_channel_sum", ptr, P, Cp, C, part.data_ptr(), out.data_ptr(), _DT[dtype], _stream())
    return out


_ZERO_POOL = {}       # device -> [optimizer epoch, zero buffer, next free element]


def _zero_grad_like(n, dev):
    """All-zero fp32 gradient for a conv bias that sits ahead of a batch-statistics BatchNorm (its gradient is
    identically zero).  Slices of one zero buffer per device and optimizer step: ONE fill per step instead of a fill or
    a copy per layer -- autograd adopts a fresh view as `.grad` without copying it (a tensor shared between parameters
    would be cloned by AccumulateGrad: 18 small copies per U-Net step).  The slices are disjoint, and nothing in the
    training path does more to a gradient than scale or reduce it, which keeps zeros zero."""
    key = torch.device(dev)
    if key.type == "cuda" and key.index is None:
        key = torch.device("cuda", torch.cuda.current_device())
    pool = _ZERO_POOL.get(key)
    if pool is None or pool[0] != _OPT_EPOCH[0] or pool[2] + n > pool[1].numel():
        pool = [_OPT_EPOCH[0], torch.zeros(max(16384, 4 * n), dtype=torch.float32, device=key), 0]
        _ZERO_POOL[key] = pool
    off = pool[2]
    pool[2] = off + (n + 3) // 4 * 4          # keep every slice 16-byte aligned
    return pool[1][off:off + n]


_NBT_DEFER = None


class defer_batch_counters:
    """Inside this context the BatchNorm `num_batches_tracked` increments of all DoubleConv blocks are collected and
    applied by ONE fused foreach kernel on exit (18 one-element kernels per U-Net step otherwise)."""

    def __enter__(self):
        global _NBT_DEFER
        self.prev, _NBT_DEFER = _NBT_DEFER, []
        return self

    def __exit__(self, *exc):
        global _NBT_DEFER
        pending, _NBT_DEFER = _NBT_DEFER, self.prev
        if pending and exc[0] is None:
            torch._foreach_add_(pending, 1)
        return False


def _bump_batch_counters(*bns):
    for bn in bns:
        if bn.num_batches_tracked is None:
            continue
        if _NBT_DEFER is not None:
            _NBT_DEFER.append(bn.num_batches_tracked)
        else:
            bn.num_batches_tracked.add_(1)


def _param_f32(p):
    t = p.detach()
    if t.dtype != torch.float32 or not t.is_contiguous():
        t = t.float().contiguous()
    return t


# ------------------------------------------------------------------------------------------------
# Building blocks shared by the autograd Functions below.  Everything a backward kernel hands to a later backward
# kernel (BatchNorm reductions accumulated by the pooling / head backward, channel sums of a concat gradient) stays
# inside ONE autograd node's forward/backward pair: nothing is hung on tensors or modules.
def _bn_momentum(bn):
    if bn.momentum is None:
        raise NotImplementedError("BatchNorm2d(momentum=None) (cumulative moving average) is not supported by the fused "
                                  "DoubleConv kernels; the reference uses the default momentum 0.1 (unet/unet.py:17,20)")
    return bn.momentum


def _convt_fwd(mod, x_t, px, w, b, B, H, W, Cin, Cout, dtype, dev, both=False):
    """ConvTranspose2d(k=2, s=2) forward as one GEMM with a pixel-shuffle store -> NHWC buffer [B,2H,2W,Coutp]."""
    Cinp, Coutp = pad32(Cin), pad32(Cout)
    if both:      # a backward will follow: both layouts, registered for the one-launch repack
        wp = mod.cache.get_pair(("tf", dtype), ("td", dtype), w, lambda: (pack_convt(w, dtype, 0), pack_convt(w, dtype, 1)),
                                meta=(1, 0, 0, dtype))
    else:
        wp = mod.cache.get(("tf", dtype), w, lambda: pack_convt(w, dtype, 0))

    def bias4():
        t = torch.zeros((4, Coutp), dtype=torch.float32, device=dev)
        t[:, :Cout] = _param_f32(b)
        return t
    b4 = None if b is None else mod.cache.get_registered(("tb", dtype), b, bias4, meta=(2, 0, 0, dtype))
    out = torch.empty((B, 2 * H, 2 * W, Coutp), dtype=dtype, device=dev)
    with _span("convt_fwd", 2.0 * B * H * W * Cin * 4 * Cout, B * H * W * (Cin + 4 * Cout) * _es(dtype)):
        _lib.call("segk_convt2x2_fwd", px, wp.data_ptr(), _p(b4), out.data_ptr(), B, H, W, Cinp, Coutp, _DT[dtype],
                  _stream())
    return out


def _convt_bwd(mod, x_t, w, pd, B, H, W, Cin, Cout, dtype, dev, need_dx, has_bias, chan_sum=None, batch=None):
    """Backward of _convt_fwd given the pointer of dout [B,2H,2W,Coutp] -> (dx act view or None, dw, db).
    chan_sum: per-channel sum of dout when the kernel that wrote dout already produced it (the bias gradient)."""
    Cinp, Coutp = pad32(Cin), pad32(Cout)
    px = act_info(x_t, dtype)[0]
    dx = None
    if need_dx:
        wd = mod.cache.get(("td", dtype), w, lambda: pack_convt(w, dtype, 1))
        dxb = torch.empty((B, H, W, Cinp), dtype=dtype, device=dev)
        with _span("convt_dgrad", 2.0 * B * H * W * Cin * 4 * Cout, B * H * W * (Cin + 4 * Cout) * _es(dtype)):
            _lib.call("segk_convt2x2_dgrad", pd, wd.data_ptr(), dxb.data_ptr(), B, H, W, Cinp, Coutp, _DT[dtype],
                      _stream())
        dx = act_view(dxb, Cin)
    slabs, S = wgrad(px, Cinp, pd, Coutp, 0, 0, B, H, W, 2, dtype, dev, alg=(Cin, Cout))
    if batch is not None:
        dw = batch.add_weight(slabs, S, w.shape, Cin, Cout, 0, 4, param=w)
    else:
        dw = wgrad_to_param(slabs, S, w.shape, Cin, Cout, 0, 4, dev, param=w)
    db = None
    if has_bias:
        db = chan_sum if chan_sum is not None else channel_sum(pd, B * 4 * H * W, Cout, dtype, dev)
    return dx, dw, db


def _head_fwd(px, Cp, w, b, B, C, H, W, dtype, dev, bn=None):
    """bn = (scale, shift): px points at the block's PRE-ACTIVATION z and the kernel forms relu(z*scale+shift) itself
    (HEAD_ON_Z: the block output is never written)."""
    ncls = w.shape[0]
    if ncls > _lib.MAX_CLASSES:
        raise RuntimeError(f"output head supports up to {_lib.MAX_CLASSES} classes, got {ncls}")
    w2 = _param_f32(w).reshape(ncls, C)
    logits = torch.empty((B, ncls, H, W), dtype=torch.float32, device=dev)
    with _span("head_fwd", 2.0 * B * H * W * C * ncls, B * H * W * (C * _es(dtype) + 4 * ncls)):
        if bn is not None:
            _lib.call("segk_head_fwd_bn", px, bn[0].data_ptr(), bn[1].data_ptr(), w2.data_ptr(), _param_f32(b).data_ptr(),
                      logits.data_ptr(), B, H, W, Cp, C, ncls, _DT[dtype], _stream())
        else:
            _lib.call("segk_head_fwd", px, w2.data_ptr(), _param_f32(b).data_ptr(), logits.data_ptr(), B, H, W, Cp, C,
                      ncls, _DT[dtype], _stream())
    return logits


def _head_bwd(dlogits, px, Cp, w, B, C, H, W, dtype, dev, bn=None, on_z=False):
    """-> (dy NHWC buffer, dw, db, bn_ready).  bn = (scale, shift, mean, rstd) of the BatchNorm whose output y the head
    reads: the kernel then also accumulates that BatchNorm's backward reductions over the dy it writes
    (bn_ready = (partials, rows) for bn_relu_bwd).  on_z: px points at the pre-activation z (see _head_fwd)."""
    ncls = w.shape[0]
    dl = dlogits
    if dl.dtype != torch.float32 or not dl.is_contiguous():
        dl = dl.float().contiguous()
    w2 = _param_f32(w).reshape(ncls, C)
    dy = torch.empty((B, H, W, Cp), dtype=dtype, device=dev)
    part = _f32(_lib.query("segk_head_part_floats", B * H * W, Cp), dev)
    dw = _grad_buffer(w, w.shape, dev)
    db = _f32(ncls, dev)
    ready = None
    with _span("head_bwd", 4.0 * B * H * W * C * ncls, B * H * W * (2 * C * _es(dtype) + 4 * ncls)):
        if on_z:
            sc, sh, mu, rs = bn
            bnpart, nb = None, 0
            if FUSE_BN_REDUCE:
                nb = _lib.query("segk_head_bwd_blocks", B * H * W)
                bnpart = _f32(nb * Cp * 2, dev)
                ready = (bnpart, nb)
            _lib.call("segk_head_bwd_bn", dl.data_ptr(), px, w2.data_ptr(), dy.data_ptr(), part.data_ptr(),
                      dw.data_ptr(), db.data_ptr(), B, H, W, Cp, C, ncls, sc.data_ptr(), sh.data_ptr(), mu.data_ptr(),
                      rs.data_ptr(), _p(bnpart), _DT[dtype], _stream())
        elif bn is not None and FUSE_BN_REDUCE:
            sc, sh, mu, rs = bn
            nb = _lib.query("segk_head_bwd_blocks", B * H * W)
            bnpart = _f32(nb * Cp * 2, dev)
            _lib.call("segk_head_bwd_bnstat", dl.data_ptr(), px, w2.data_ptr(), dy.data_ptr(), part.data_ptr(),
                      dw.data_ptr(), db.data_ptr(), B, H, W, Cp, C, ncls, sc.data_ptr(), sh.data_ptr(), mu.data_ptr(),
                      rs.data_ptr(), bnpart.data_ptr(), _DT[dtype], _stream())
            ready = (bnpart, nb)
        else:
            _lib.call("segk_head_bwd", dl.data_ptr(), px, w2.data_ptr(), dy.data_ptr(), part.data_ptr(), dw.data_ptr(),
                      db.data_ptr(), B, H, W, Cp, C, ncls, _DT[dtype], _stream())
    return dy, dw, db, ready


class BlockCfg:
    """Static (non-tensor) configuration of one DoubleConvFn call."""
    __slots__ = ("mod", "up_mod", "emit_pool")

    def __init__(self, mod, up_mod=None, emit_pool=False):
        self.mod, self.up_mod, self.emit_pool = mod, up_mod, emit_pool


class DoubleConvFn(torch.autograd.Function):
    """One autograd node per reference block:

        [u = ConvTranspose2d(up_x)]                       reference unet/unet.py:59,62-63 (Up: xb = u, concat [xa | u])
        y = ReLU(BN2(Conv3x3(ReLU(BN1(Conv3x3([xa | xb]))))))   unet/unet.py:4-25 (bias=True), clip/clipunet.py:86-93
        [pooled = MaxPool2d(2,2)(y)]                      unet/unet.py:40 (the next Down block's pooling)
        [logits = Conv2d(C, classes, 1)(y)]               unet/unet.py:91,105; clip/clipunet.py:181,187

    xb is the optional second concat operand, consumed without materialising the concat.  The optional stages live in
    the same node because their backward kernels feed each other: the pooling / head backward writes the complete
    gradient of y and accumulates BN2's backward reductions on the way (the block's backward then skips its reduce pass
    over dy and z2), and the concat data-gradient's per-channel sums are the ConvTranspose bias gradient.

    Forward kernels: conv3x3(+stats) -> bn_finalize -> conv3x3 with BN1+ReLU fused in its load prologue (+stats) ->
    bn_finalize -> bn_relu_apply (+pool).  The conv biases never enter the kernels: ahead of a training-mode BatchNorm
    they cancel (they only shift running_mean, handled in bn_finalize), and in eval mode they fold into the BN shift.

    Returns y, or (y, pooled) with cfg.emit_pool, or the fp32 NCHW logits when head_w is given.
    """

    @staticmethod
    def forward(ctx, cfg, xa, xb, w1, b1, g1, be1, w2, b2, g2, be2, up_x, up_w, up_b, head_w, head_b):
        with _level_scope(cfg.mod, False):
            return DoubleConvFn._forward(ctx, cfg, xa, xb, w1, b1, g1, be1, w2, b2, g2, be2, up_x, up_w, up_b, head_w, head_b)

    @staticmethod
    def _forward(ctx, cfg, xa, xb, w1, b1, g1, be1, w2, b2, g2, be2, up_x, up_w, up_b, head_w, head_b):
        mod = cfg.mod
        dtype = mod.compute_dtype or _compute_dtype
        _require_cuda(xa, "DoubleConvReLU")
        dev = xa.device
        B, CA, H, W = xa.shape
        # ---- optional ConvTranspose2d producing the second concat operand
        up_t, up_dims = None, None
        if up_x is not None:
            if xb is not None:
                raise ValueError("DoubleConvFn: give either xb or up_x")
            _require_cuda(up_x, "ConvTranspose2d")
            Bu, Cin_u, Hu, Wu = up_x.shape
            Cout_u = up_w.shape[1]
            if (2 * Hu, 2 * Wu) != (H, W):
                raise RuntimeError(f"Sizes of tensors must match except in dimension 1: {tuple(xa.shape)} vs "
                                   f"{(Bu, Cout_u, 2 * Hu, 2 * Wu)} (H and W must be multiples of 16)")
            up_t, pux, _ = _raw(up_x, dtype)
            ubuf = _convt_fwd(cfg.up_mod, up_t, pux, up_w, up_b, Bu, Hu, Wu, Cin_u, Cout_u, dtype, dev,
                              both=any(ctx.needs_input_grad))
            xb = act_view(ubuf, Cout_u)
            up_dims = (Bu, Hu, Wu, Cin_u, Cout_u)
        CB = 0 if xb is None else xb.shape[1]
        Cout = w1.shape[0]
        Coutp = pad32(Cout)
        training = mod.training
        bn1, bn2 = mod.bn_modules()
        want_grad = any(ctx.needs_input_grad)
        # the stem (reference unet.py:16 on the NCHW fp32 batch of training.py:45): a kernel of its own gathers the im2col
        # operand straight from the image -- no layout pass, no 32-channel padding in the matrix work (segk_stem3x3)
        stem_rows = 0
        if (xb is None and act_info(xa, dtype) is None and xa.dtype == torch.float32 and xa.is_contiguous()
                and w1.dtype == torch.float32 and w1.is_contiguous()):
            stem_rows = _lib.query("segk_stem3x3_rows", B, H, W, CA, Cout, _DT[dtype])
        stem_raw = False
        if stem_rows:
            xa_t, pA, CAp = None, 0, pad32(CA)
            if want_grad and _lib.query("segk_stem3x3_wgrad_slabs", B, H, W, CA, Cout, _DT[dtype]):
                xa_t, stem_raw = xa.detach(), True   # the weight gradient gathers from the NCHW batch too: nothing to keep
            elif want_grad:                          # the weight gradient reads the padded NHWC copy: written on the way
                xbuf = torch.empty((B, H, W, CAp), dtype=dtype, device=dev)
                xa_t, pA = act_view(xbuf, CA), xbuf.data_ptr()
        else:
            xa_t, pA, CAp = _raw(xa, dtype)
        xb_t, pB, CBp = (None, 0, 0) if xb is None else _raw(xb, dtype)
        if training and want_grad:     # a backward will follow: both layouts in one pass per weight
            w1p = mod.cache.get_pair(("w1f", dtype), ("w1d", dtype), w1, lambda: pack_conv_both(w1, CA, CB, dtype),
                                     meta=(0, CA, CB, dtype))
            w2p = mod.cache.get_pair(("w2f", dtype), ("w2d", dtype), w2, lambda: pack_conv_both(w2, Cout, 0, dtype),
                                     meta=(0, Cout, 0, dtype))
        else:
            w1p = mod.cache.get(("w1f", dtype), w1, lambda: pack_conv(w1, CA, CB, dtype, 0))
            w2p = mod.cache.get(("w2f", dtype), w2, lambda: pack_conv(w2, Cout, 0, dtype, 0))
        tiles1 = stem_rows or _lib.query("segk_conv_tiles", B, H, W, CAp + CBp, Coutp, _DT[dtype])
        tiles2 = _lib.query("segk_conv_tiles", B, H, W, Coutp, Coutp, _DT[dtype])
        P = B * H * W

        z1 = torch.empty((B, H, W, Coutp), dtype=dtype, device=dev)
        st1 = _f32(_lib.query("segk_bn_stats_floats", tiles1, Coutp), dev) if training else None
        if stem_rows:
            e = _es(dtype)
            with _span("conv3x3_igemm", 2.0 * P * 9 * CA * Cout, P * (4 * CA + Cout * e) + 9.0 * CA * Cout * 4):
                _lib.call("segk_stem3x3", xa.data_ptr(), w1.data_ptr(), z1.data_ptr(), pA, _p(st1), B, H, W, CA, Cout,
                          _DT[dtype], _stream())
        else:
            conv3x3(xa_t, pA, CAp, pB, CBp, w1p, z1.data_ptr(), Coutp, 0, 0, B, H, W, dtype, stats=st1, alg=(CA + CB, Cout))
        sc1, sh1, mu1, rs1 = bn_finalize(st1, tiles1, Cout, P, None if b1 is None else _param_f32(b1), _param_f32(g1),
                                         _param_f32(be1), bn1.running_mean, bn1.running_var, _bn_momentum(bn1), bn1.eps,
                                         training, dev)
        z2 = torch.empty((B, H, W, Coutp), dtype=dtype, device=dev)
        st2 = _f32(_lib.query("segk_bn_stats_floats", tiles2, Coutp), dev) if training else None
        # training: where the layer's kernel supports it, conv2's BN+ReLU prologue also writes the hidden activation
        # a1 = relu(bn1(z1)) it computes on the fly, so the weight-gradient pass reads a1 instead of re-deriving it
        # from z1 fragment by fragment (measured 25-30 % of that kernel)
        a1 = None
        if training and want_grad and _lib.query("segk_conv_writes_act_q", Coutp, Coutp, _DT[dtype]):
            a1 = torch.empty((B, H, W, Coutp), dtype=dtype, device=dev)
            with _span("conv3x3_igemm", 2.0 * P * 9 * Cout * Cout, (P * 2 * Cout + 9.0 * Cout * Cout) * _es(dtype)):
                _lib.call("segk_conv3x3_act", z1.data_ptr(), w2p.data_ptr(), sc1.data_ptr(), sh1.data_ptr(), z2.data_ptr(),
                          a1.data_ptr(), _p(st2), B, H, W, Coutp, Coutp, _DT[dtype], _stream())
        else:
            conv3x3(z1, z1.data_ptr(), Coutp, 0, 0, w2p, z2.data_ptr(), Coutp, 0, 0, B, H, W, dtype, scale=sc1,
                    shift=sh1, stats=st2, alg=(Cout, Cout))
        sc2, sh2, mu2, rs2 = bn_finalize(st2, tiles2, Cout, P, None if b2 is None else _param_f32(b2), _param_f32(g2),
                                         _param_f32(be2), bn2.running_mean, bn2.running_var, _bn_momentum(bn2), bn2.eps,
                                         training, dev)
        if training:
            _bump_batch_counters(bn1, bn2)
        head_on_z = head_w is not None and HEAD_ON_Z
        y = None if head_on_z else torch.empty((B, H, W, Coutp), dtype=dtype, device=dev)
        pooled = None
        if head_on_z:
            pass         # the head kernels form relu(bn(z2)) themselves: the block output is never written
        elif cfg.emit_pool:
            if H < 2 or W < 2:
                raise RuntimeError("MaxPool2d(2,2): input smaller than the window")
            # this block's output feeds a Down block: BN+ReLU and its MaxPool2d(2,2) in one pass over z2
            pooled = torch.empty((B, H // 2, W // 2, Coutp), dtype=dtype, device=dev)
            with _span("bn_relu_apply", 0.0, 2.25 * P * Cout * _es(dtype)):
                _lib.call("segk_bn_relu_apply_pool", z2.data_ptr(), y.data_ptr(), pooled.data_ptr(), sc2.data_ptr(),
                          sh2.data_ptr(), B, H, W, Coutp, _DT[dtype], _stream())
        else:
            with _span("bn_relu_apply", 0.0, 2.0 * P * Cout * _es(dtype)):
                _lib.call("segk_bn_relu_apply", z2.data_ptr(), y.data_ptr(), sc2.data_ptr(), sh2.data_ptr(), P, Coutp,
                          _DT[dtype], _stream())
        logits = None
        if head_on_z:
            logits = _head_fwd(z2.data_ptr(), Coutp, head_w, head_b, B, Cout, H, W, dtype, dev, bn=(sc2, sh2))
        elif head_w is not None:
            logits = _head_fwd(y.data_ptr(), Coutp, head_w, head_b, B, Cout, H, W, dtype, dev)

        ctx.cfg, ctx.dtype, ctx.dims = cfg, dtype, (B, H, W, CA, CB, Cout)
        ctx.training = training
        ctx.has_bias = (b1 is not None, b2 is not None)
        ctx.up_dims = up_dims
        ctx.has_head = head_w is not None
        ctx.head_on_z = head_on_z
        ctx.stem_raw = stem_raw
        ctx.save_for_backward(xa_t, xb_t, z1, z2, sc1, sh1, mu1, rs1, sc2, sh2, mu2, rs2, w1, w2, a1,
                              y if (cfg.emit_pool or head_w is not None) else None, up_t, up_w, head_w)
        if logits is not None:
            return logits
        if pooled is not None:
            y_out = act_view(y, Cout)
            ctx.y_ref = weakref.ref(y_out)     # backward asks it whether the caller hooked the skip tensor (see there)
            return y_out, act_view(pooled, Cout)
        return act_view(y, Cout)

    @staticmethod
    def backward(ctx, *grads):
        with _level_scope(ctx.cfg.mod, True):
            return DoubleConvFn._backward(ctx, *grads)

    @staticmethod
    def _backward(ctx, *grads):
        frozen = not ctx.training      # eval mode: both BatchNorms are fixed affine maps (running statistics)
        (xa_t, xb_t, z1, z2, sc1, sh1, mu1, rs1, sc2, sh2, mu2, rs2, w1, w2, a1, y, up_t, up_w, head_w) = ctx.saved_tensors
        cfg, dtype = ctx.cfg, ctx.dtype
        mod = cfg.mod
        B, H, W, CA, CB, Cout = ctx.dims
        dev = z1.device
        Coutp, CAp, CBp = pad32(Cout), pad32(CA), (pad32(CB) if CB else 0)
        P = B * H * W
        pA = 0 if ctx.stem_raw else act_info(xa_t, dtype)[0]
        pB = 0 if xb_t is None else act_info(xb_t, dtype)[0]
        need = ctx.needs_input_grad

        # ---- gradient of the block output y, complete; `ready` = BN2's backward reductions when the kernel that wrote
        # it accumulated them on the way
        ready = None
        d_head_w = d_head_b = None
        keep = None
        if ctx.has_head:
            src = z2 if ctx.head_on_z else y
            dy_buf, d_head_w, d_head_b, ready = _head_bwd(grads[0], src.data_ptr(), Coutp, head_w, B, Cout, H, W, dtype, dev,
                                                          bn=(sc2, sh2, mu2, rs2), on_z=ctx.head_on_z)
            pdy, keep = dy_buf.data_ptr(), dy_buf
        elif cfg.emit_pool:
            dy, dpool = grads
            rc_skip, rc_pool = sys.getrefcount(dy), sys.getrefcount(dpool)   # before anything below takes references
            if dpool is None:
                keep, pdy, _ = _raw(dy, dtype)
            else:
                dp_t, pdp, _ = _raw(dpool, dtype)
                if dy is None:
                    keep = torch.empty((B, H, W, Coutp), dtype=dtype, device=dev)
                    pdy, acc, nbytes = keep.data_ptr(), 0, 2.25
                else:
                    # the skip gradient arrives as an act tensor (fresh output of the consumer's data-gradient kernel;
                    # converted copy otherwise): the pooled gradient is routed INTO its storage, one kernel instead of
                    # a pooling backward plus a full-resolution add.  A backward must not change a gradient somebody
                    # else can still see: with a tensor hook on the skip tensor (register_hook: the hook may keep the
                    # tensor it is handed; retain_grad() clones and is safe) the sum goes to a private copy instead.
                    # Autograd itself can hold it too: torch.autograd.grad(loss, [skip]) CAPTURES the very tensor this
                    # backward is handed (no clone).  Every other holder shows in the tensor's reference counts -- the
                    # engine's input buffer and this frame account for a TensorImpl use count of 2, a capture makes it 3;
                    # a hook that outlived the Python skip tensor and stashed its argument raises the Python reference
                    # count above that of the sibling gradient that arrived in the same tuple.
                    yo = getattr(ctx, "y_ref", lambda: None)()
                    uc = getattr(dy, "_use_count", None)
                    shared = uc is None or uc() > 2 or rc_skip > rc_pool
                    hooked = INPLACE_SKIP_GRAD is False or shared or (yo is not None and bool(yo._backward_hooks))
                    keep, pdy, _ = _raw(dy, dtype)
                    if hooked and keep.data_ptr() == dy.data_ptr():      # zero-copy act tensor: the caller's own storage
                        keep = _private_act_copy(keep, dtype)
                        pdy = keep.data_ptr()
                    acc, nbytes = 1, 3.25
                nb = _lib.query("segk_maxpool_bwd_stat_blocks", B, H, W, Coutp, _DT[dtype]) if FUSE_BN_REDUCE else 0
                with _span("maxpool_bwd", 0.0, nbytes * P * Cout * _es(dtype)):
                    if nb > 0:
                        part = _f32(nb * Coutp * 2, dev)
                        _lib.call("segk_maxpool2x2_bwd_bnstat", y.data_ptr(), pdp, pdy, B, H, W, Coutp, acc,
                                  sc2.data_ptr(), sh2.data_ptr(), mu2.data_ptr(), rs2.data_ptr(), part.data_ptr(),
                                  z2.data_ptr(), _DT[dtype], _stream())
                        ready = (part, nb)
                    else:
                        _lib.call("segk_maxpool2x2_bwd", y.data_ptr(), pdp, pdy, B, H, W, Coutp, acc, _DT[dtype],
                                  _stream())
        else:
            keep, pdy, _ = _raw(grads[0], dtype)

        # ---- second conv: BN2+ReLU backward, data gradient, weight gradient
        dz2 = torch.empty((B, H, W, Coutp), dtype=dtype, device=dev)
        dg2, dbe2 = bn_relu_bwd(pdy, z2.data_ptr(), dz2.data_ptr(), sc2, sh2, mu2, rs2, P, Cout, dtype, dev, ready=ready,
                                frozen=frozen)
        del keep
        w2d = mod.cache.get(("w2d", dtype), w2, lambda: pack_conv(w2, Cout, 0, dtype, 1))
        da1 = torch.empty((B, H, W, Coutp), dtype=dtype, device=dev)
        conv3x3(dz2, dz2.data_ptr(), Coutp, 0, 0, w2d, da1.data_ptr(), Coutp, 0, 0, B, H, W, dtype, alg=(Cout, Cout))
        if a1 is not None:
            slabs, S = wgrad(dz2.data_ptr(), Coutp, a1.data_ptr(), Coutp, 0, 0, B, H, W, 0, dtype, dev, alg=(Cout, Cout))
        else:
            slabs, S = wgrad(dz2.data_ptr(), Coutp, z1.data_ptr(), Coutp, 0, 0, B, H, W, 0, dtype, dev, scale=sc1,
                             shift=sh1, alg=(Cout, Cout))
        # (each slab reduction runs right behind its weight-gradient kernel, while the slabs still sit in the Infinity
        # Cache: deferring a block's reductions into one launch at the end read them back from HBM and was slower)
        dw2 = wgrad_to_param(slabs, S, w2.shape, Cout, Cout, 0, 9, dev, param=w2)
        rb = ReduceBatch(dev)        # the ConvTranspose weight gradient and its bias gradient's column sums: one launch
        del slabs, dz2

        # ---- first conv (dz1 overwrites da1 in place: da1 is private to this function)
        dg1, dbe1 = bn_relu_bwd(da1.data_ptr(), z1.data_ptr(), da1.data_ptr(), sc1, sh1, mu1, rs1, P, Cout, dtype, dev,
                                frozen=frozen)
        dz1 = da1
        dxa = dxb = None
        dxb_buf, chan_sum = None, None
        has_up = ctx.up_dims is not None
        if need[1] or need[2] or has_up:
            w1d = mod.cache.get(("w1d", dtype), w1, lambda: pack_conv(w1, CA, CB, dtype, 1))
            dxa_buf = torch.empty((B, H, W, CAp), dtype=dtype, device=dev)
            dxb_buf = torch.empty((B, H, W, CBp), dtype=dtype, device=dev) if CB else None
            # with the ConvTranspose output as concat operand the kernel's per-channel-sum epilogue also yields
            # sum_pixels(dxb): exactly the ConvTranspose bias gradient, without another pass over dxb
            std = None
            if has_up:
                tiles_d = _lib.query("segk_conv_tiles", B, H, W, Coutp, CAp + CBp, _DT[dtype])
                std = _f32(_lib.query("segk_bn_stats_floats", tiles_d, CAp + CBp), dev)
            conv3x3(dz1, dz1.data_ptr(), Coutp, 0, 0, w1d, dxa_buf.data_ptr(), CAp, _p(dxb_buf), CBp, B, H, W, dtype,
                    stats=std, alg=(Cout, CA + CB))
            dxa = act_view(dxa_buf, CA)
            dxb = act_view(dxb_buf, CB) if CB else None
            if has_up:
                chan_sum = rb.add_colsum(std, tiles_d, CAp + CBp, CAp, CB)
        if ctx.stem_raw:        # the stem: im2col gather from the NCHW fp32 batch, slabs in OIHW column order (k = ci*9 + tap)
            S = _lib.query("segk_stem3x3_wgrad_slabs", B, H, W, CA, Cout, _DT[dtype])
            slabs = _f32(S * 64 * 32, dev)
            with _span("wgrad3x3", 2.0 * P * 9 * CA * Cout, P * (Cout * _es(dtype) + 4 * CA) + 4.0 * 9 * CA * Cout):
                _lib.call("segk_stem3x3_wgrad", xa_t.data_ptr(), dz1.data_ptr(), slabs.data_ptr(), B, H, W, CA, Cout,
                          _DT[dtype], _stream())
            dw1 = wgrad_to_param(slabs, S, w1.shape, Cout, 9 * CA, 0, 1, dev, param=w1)
        else:
            slabs, S = wgrad(dz1.data_ptr(), Coutp, pA, CAp, pB, CBp, B, H, W, 0, dtype, dev, alg=(Cout, CA + CB))
            dw1 = wgrad_to_param(slabs, S, w1.shape, Cout, CA, CB, 9, dev, param=w1)
        del slabs
        # conv biases ahead of a batch-statistics BatchNorm have an identically zero gradient; ahead of a frozen one the
        # bias is part of the affine map: d(bias) = sum(dz) = scale * sum(g)
        if frozen:
            db1 = sc1[:Cout] * dbe1 if ctx.has_bias[0] else None
            db2 = sc2[:Cout] * dbe2 if ctx.has_bias[1] else None
        else:
            db1 = _zero_grad_like(Cout, dev) if ctx.has_bias[0] else None
            db2 = _zero_grad_like(Cout, dev) if ctx.has_bias[1] else None

        d_up_x = d_up_w = d_up_b = None
        if has_up:
            Bu, Hu, Wu, Cin_u, Cout_u = ctx.up_dims
            d_up_x, d_up_w, d_up_b = _convt_bwd(cfg.up_mod, up_t, up_w, dxb_buf.data_ptr(), Bu, Hu, Wu, Cin_u, Cout_u, dtype,
                                                dev, need[11], cfg.up_mod.upsample.bias is not None, chan_sum=chan_sum,
                                                batch=rb)
            dxb = None                   # xb was internal
        rb.flush()
        return (None, dxa, dxb, dw1, db1, dg1, dbe1, dw2, db2, dg2, dbe2, d_up_x, d_up_w, d_up_b, d_head_w, d_head_b)


def double_conv(mod, xa, xb, params, emit_pool=False, head=None, up_mod=None, up_x=None):
    """Module-side entry: params = (w1, b1, g1, be1, w2, b2, g2, be2); head = nn.Conv2d(C, classes, 1) or None;
    up_mod = the module owning `upsample` (nn.ConvTranspose2d) whose output is the second concat operand."""
    cfg = BlockCfg(mod, up_mod, emit_pool and head is None)
    uw = ub = None
    if up_mod is not None:
        uw, ub = up_mod.upsample.weight, up_mod.upsample.bias
    hw = hb = None
    if head is not None:
        hw, hb = head.weight, head.bias
    return DoubleConvFn.apply(cfg, xa, xb, *params, up_x, uw, ub, hw, hb)


class MaxPoolFn(torch.autograd.Function):
    """nn.MaxPool2d(2,2) -- reference unet/unet.py:40."""

    @staticmethod
    def forward(ctx, x, dtype):
        _require_cuda(x, "MaxPool2d")
        x_t, px, Cp = _raw(x, dtype)
        B, C, H, W = x.shape
        y = torch.empty((B, H // 2, W // 2, Cp), dtype=dtype, device=x.device)
        with _span("maxpool_fwd", 0.0, 1.25 * B * H * W * C * _es(dtype)):
            _lib.call("segk_maxpool2x2_fwd", px, y.data_ptr(), B, H, W, Cp, _DT[dtype], _stream())
        ctx.save_for_backward(x_t)
        ctx.dtype = dtype
        return act_view(y, C)

    @staticmethod
    def backward(ctx, dy):
        (x_t,) = ctx.saved_tensors
        dtype = ctx.dtype
        B, C, H, W = x_t.shape
        px, Cp = act_info(x_t, dtype)
        dy_t, pdy, _ = _raw(dy, dtype)
        dx = torch.empty((B, H, W, Cp), dtype=dtype, device=x_t.device)
        with _span("maxpool_bwd", 0.0, 2.25 * B * H * W * C * _es(dtype)):
            _lib.call("segk_maxpool2x2_bwd", px, pdy, dx.data_ptr(), B, H, W, Cp, 0, _DT[dtype], _stream())
        return act_view(dx, C), None


class ConvT2x2Fn(torch.autograd.Function):
    """nn.ConvTranspose2d(Cin, Cout, kernel_size=2, stride=2) -- reference unet/unet.py:59, clip/clipunet.py:83.
    Non-overlapping, so forward is one GEMM [P x Cin].[Cin x 4Cout] with a pixel-shuffle store."""

    @staticmethod
    def forward(ctx, mod, x, w, b):
        dtype = mod.compute_dtype or _compute_dtype
        _require_cuda(x, "ConvTranspose2d")
        B, Cin, H, W = x.shape
        Cout = w.shape[1]
        x_t, px, _ = _raw(x, dtype)
        out = _convt_fwd(mod, x_t, px, w, b, B, H, W, Cin, Cout, dtype, x.device, both=any(ctx.needs_input_grad))
        ctx.mod, ctx.dtype, ctx.dims = mod, dtype, (B, H, W, Cin, Cout)
        ctx.has_bias = b is not None
        ctx.save_for_backward(x_t, w)
        return act_view(out, Cout)

    @staticmethod
    def backward(ctx, dout):
        x_t, w = ctx.saved_tensors
        B, H, W, Cin, Cout = ctx.dims
        d_t, pd, _ = _raw(dout, ctx.dtype)
        dx, dw, db = _convt_bwd(ctx.mod, x_t, w, pd, B, H, W, Cin, Cout, ctx.dtype, x_t.device, ctx.needs_input_grad[1],
                                ctx.has_bias)
        return None, dx, dw, db


class Conv1x1Fn(torch.autograd.Function):
    """nn.Conv2d(Cin, Cout, kernel_size=1) between act tensors -- reference clip/clipunet.py:84,122."""

    @staticmethod
    def forward(ctx, mod, x, w, b):
        dtype = mod.compute_dtype or _compute_dtype
        _require_cuda(x, "Conv2d 1x1")
        dev = x.device
        B, Cin, H, W = x.shape
        Cout = w.shape[0]
        Cinp, Coutp = pad32(Cin), pad32(Cout)
        x_t, px, _ = _raw(x, dtype)
        # both operands are registered for the model's one-launch re-pack after an optimizer step (repack_stale)
        wp = mod.cache.get_registered(("cf", dtype), w, lambda: pack_conv(w, Cin, 0, dtype, 0, taps=1), meta=(3, Cin, 0, dtype))

        def biasp():
            t = torch.zeros((Coutp,), dtype=torch.float32, device=dev)
            t[:Cout] = _param_f32(b)
            return t
        bp = None if b is None else mod.cache.get_registered(("cb", dtype), b, biasp, meta=(2, 1, 0, dtype))
        out = torch.empty((B, H, W, Coutp), dtype=dtype, device=dev)
        _lib.call("segk_conv1x1", px, wp.data_ptr(), _p(bp), out.data_ptr(), B, H, W, Cinp, Coutp, _DT[dtype], _stream())
        ctx.mod, ctx.dtype, ctx.dims = mod, dtype, (B, H, W, Cin, Cout)
        ctx.has_bias = b is not None
        ctx.save_for_backward(x_t, w)
        return act_view(out, Cout)

    @staticmethod
    def backward(ctx, dout):
        x_t, w = ctx.saved_tensors
        mod, dtype = ctx.mod, ctx.dtype
        B, H, W, Cin, Cout = ctx.dims
        dev = x_t.device
        Cinp, Coutp = pad32(Cin), pad32(Cout)
        d_t, pd, _ = _raw(dout, dtype)
        px = act_info(x_t, dtype)[0]
        dx = None
        if ctx.needs_input_grad[1]:
            wd = mod.cache.get(("cd", dtype), w, lambda: pack_conv(w, Cin, 0, dtype, 1, taps=1))
            dxb = torch.empty((B, H, W, Cinp), dtype=dtype, device=dev)
            _lib.call("segk_conv1x1", pd, wd.data_ptr(), 0, dxb.data_ptr(), B, H, W, Coutp, Cinp, _DT[dtype], _stream())
            dx = act_view(dxb, Cin)
        slabs, S = wgrad(pd, Coutp, px, Cinp, 0, 0, B, H, W, 1, dtype, dev)
        dw = wgrad_to_param(slabs, S, w.shape, Cout, Cin, 0, 1, dev, param=w)
        db = channel_sum(pd, B * H * W, Cout, dtype, dev) if ctx.has_bias else None
        return None, dx, dw, db


class Conv3x3Fn(torch.autograd.Function):
    """Plain nn.Conv2d(Cin, Cout, kernel_size=3, padding=1) with bias and no BatchNorm behind it -- the
    reconstruction head of reference autoencoder/autoencoder.py:188-191 (the Sigmoid stays a stock torch op on the
    small fp32 output).  Returns an act tensor."""

    @staticmethod
    def forward(ctx, mod, x, w, b):
        dtype = mod.compute_dtype or _compute_dtype
        _require_cuda(x, "Conv2d 3x3")
        dev = x.device
        B, Cin, H, W = x.shape
        Cout = w.shape[0]
        Cinp, Coutp = pad32(Cin), pad32(Cout)
        x_t, px, _ = _raw(x, dtype)
        wp = mod.cache.get(("c3f", dtype), w, lambda: pack_conv(w, Cin, 0, dtype, 0))

        def biasp():
            t = torch.zeros((Coutp,), dtype=torch.float32, device=dev)
            t[:Cout] = _param_f32(b)
            return t
        bp = None if b is None else mod.cache.get(("c3b", dtype), b, biasp)
        out = torch.empty((B, H, W, Coutp), dtype=dtype, device=dev)
        P, e = B * H * W, _es(dtype)
        with _span("conv3x3_igemm", 2.0 * P * 9 * Cin * Cout, P * (Cin + Cout) * e + 9.0 * Cin * Cout * e):
            _lib.call("segk_conv3x3", px, 0, wp.data_ptr(), _p(bp), 0, 0, out.data_ptr(), 0, 0, B, H, W, Cinp, 0, Coutp, 0,
                      _DT[dtype], _stream())
        ctx.mod, ctx.dtype, ctx.dims = mod, dtype, (B, H, W, Cin, Cout)
        ctx.has_bias = b is not None
        ctx.save_for_backward(x_t, w)
        return act_view(out, Cout)

    @staticmethod
    def backward(ctx, dout):
        x_t, w = ctx.saved_tensors
        mod, dtype = ctx.mod, ctx.dtype
        B, H, W, Cin, Cout = ctx.dims
        dev = x_t.device
        Cinp, Coutp = pad32(Cin), pad32(Cout)
        d_t, pd, _ = _raw(dout, dtype)
        px = act_info(x_t, dtype)[0]
        dx = None
        if ctx.needs_input_grad[1]:
            wd = mod.cache.get(("c3d", dtype), w, lambda: pack_conv(w, Cin, 0, dtype, 1))
            dxb = torch.empty((B, H, W, Cinp), dtype=dtype, device=dev)
            conv3x3(d_t, pd, Coutp, 0, 0, wd, dxb.data_ptr(), Cinp, 0, 0, B, H, W, dtype, alg=(Cout, Cin))
            dx = act_view(dxb, Cin)
        slabs, S = wgrad(pd, Coutp, px, Cinp, 0, 0, B, H, W, 0, dtype, dev, alg=(Cout, Cin))
        dw = wgrad_to_param(slabs, S, w.shape, Cout, Cin, 0, 9, dev, param=w)
        db = channel_sum(pd, B * H * W, Cout, dtype, dev) if ctx.has_bias else None
        return None, dx, dw, db


class BilinearFn(torch.autograd.Function):
    """F.interpolate(x, size, mode='bilinear', align_corners=False) -- reference clip/clipunet.py:99-100."""

    @staticmethod
    def forward(ctx, x, size, dtype):
        _require_cuda(x, "bilinear resize")
        x_t, px, Cp = _raw(x, dtype)
        B, C, IH, IW = x.shape
        OH, OW = int(size[0]), int(size[1])
        y = torch.empty((B, OH, OW, Cp), dtype=dtype, device=x.device)
        with _span("bilinear_fwd", 0.0, B * (IH * IW + OH * OW) * C * _es(dtype)):
            _lib.call("segk_bilinear_fwd", px, y.data_ptr(), B, IH, IW, OH, OW, Cp, _DT[dtype], _stream())
        ctx.cfg = (B, C, IH, IW, OH, OW, Cp, dtype)
        return act_view(y, C)

    @staticmethod
    def backward(ctx, dy):
        B, C, IH, IW, OH, OW, Cp, dtype = ctx.cfg
        d_t, pd, _ = _raw(dy, dtype)
        dx = torch.empty((B, IH, IW, Cp), dtype=dtype, device=dy.device)
        with _span("bilinear_bwd", 0.0, B * (IH * IW + OH * OW) * C * _es(dtype)):
            # separable two-pass form when the map grows by more than 2x (the CLIP skips: 14 -> 28..224)
            tmp = _f32(B * OH * IW * Cp, dy.device) if OH * OW > 4 * IH * IW else None
            _lib.call("segk_bilinear_bwd", pd, dx.data_ptr(), _p(tmp), B, IH, IW, OH, OW, Cp, _DT[dtype], _stream())
        return act_view(dx, C), None, None


class HeadFn(torch.autograd.Function):
    """Output nn.Conv2d(C, num_classes, 1) -- reference unet/unet.py:91,105; clip/clipunet.py:181,187.
    Returns fp32 NCHW logits exactly like the reference module does.  (Behind a DoubleConv block the models pass the
    head INTO that block's autograd node instead -- ops.double_conv(head=...) -- so that the head backward can accumulate
    the block's BatchNorm reductions; this stand-alone form is the plain layer.)"""

    @staticmethod
    def forward(ctx, mod, x, w, b):
        dtype = mod.compute_dtype or _compute_dtype
        _require_cuda(x, "output Conv2d")
        B, C, H, W = x.shape
        x_t, px, Cp = _raw(x, dtype)
        logits = _head_fwd(px, Cp, w, b, B, C, H, W, dtype, x.device)
        ctx.dtype, ctx.dims = dtype, (B, C, H, W)
        ctx.save_for_backward(x_t, w)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        x_t, w = ctx.saved_tensors
        dtype = ctx.dtype
        B, C, H, W = ctx.dims
        px, Cp = act_info(x_t, dtype)
        dy, dw, db, _ = _head_bwd(dlogits, px, Cp, w, B, C, H, W, dtype, x_t.device)
        return None, act_view(dy, C), dw, db


class SegLossFn(torch.autograd.Function):
    """dice_weight * softDice + ce_weight * CrossEntropy in one pass over the logits --
    reference utils/weighted_loss.py:31-98,140-166 and nn.CrossEntropyLoss as called at utils/training.py:47."""

    @staticmethod
    def forward(ctx, logits, target, class_weights, ignore_index, smooth, dice_weight, ce_weight):
        _require_cuda(logits, "segmentation loss")
        if logits.dim() != 4:
            raise ValueError(f"expected logits [N,C,H,W], got {tuple(logits.shape)}")
        N, C, H, W = logits.shape
        if C > _lib.MAX_CLASSES:
            raise RuntimeError(f"fused loss supports up to {_lib.MAX_CLASSES} classes, got {C}")
        lg = logits.detach()
        if lg.dtype != torch.float32 or not lg.is_contiguous():
            lg = lg.float().contiguous()
        tg = target.detach()
        if tg.dtype != torch.int64 or not tg.is_contiguous():
            tg = tg.long().contiguous()
        if tg.numel() != N * H * W:
            raise ValueError(f"target shape {tuple(target.shape)} does not match logits {tuple(logits.shape)}")
        dev = logits.device
        cw = None
        if class_weights is not None:
            cw = class_weights.detach().to(device=dev, dtype=torch.float32).contiguous()
            if cw.numel() != C:
                raise ValueError("class_weights must have one entry per class")
        ign = -1 if ignore_index is None else int(ignore_index)
        part = _f32(_lib.query("segk_loss_part_floats", N * H * W), dev)
        state = _f32(_lib.query("segk_loss_state_floats"), dev)
        out = _f32(1, dev)          # the returned loss lives in a buffer of its own (no copy kernel, no view of saved state)
        with _span("loss_fwd", 0.0, N * H * W * (4.0 * C + 8)):
            _lib.call("segk_loss_fwd", lg.data_ptr(), tg.data_ptr(), _p(cw), N, C, H * W, ign, float(smooth),
                      float(dice_weight), float(ce_weight), part.data_ptr(), state.data_ptr(), out.data_ptr(), _stream())
        ctx.cfg = (N, C, H, W, ign, float(dice_weight), float(ce_weight))
        ctx.save_for_backward(lg, tg, cw, state)
        return out.view(())

    @staticmethod
    def backward(ctx, gout):
        lg, tg, cw, state = ctx.saved_tensors
        N, C, H, W, ign, dw, cew = ctx.cfg
        go = gout.detach().float().reshape(1).contiguous()
        dl = torch.empty_like(lg)
        with _span("loss_bwd", 0.0, N * H * W * (8.0 * C + 8)):
            _lib.call("segk_loss_bwd", lg.data_ptr(), tg.data_ptr(), _p(cw), state.data_ptr(), go.data_ptr(), N, C,
                      H * W, ign, dw, cew, dl.data_ptr(), _stream())
        return dl, None, None, None, None, None, None


class ProbLossFn(torch.autograd.Function):
    """dice_weight * softDice(p) + nll_weight * NLLLoss(log(p + eps) or p) on class PROBABILITIES (no softmax inside) --
    reference utils/weighted_loss.py:170-343 as prompt_based/prompt.ipynb configures it (apply_softmax=False,
    nll_nonlin = log(x + 1e-9))."""

    @staticmethod
    def forward(ctx, probs, target, class_weights, ignore_index, smooth, dice_weight, nll_weight, nll_log, eps):
        _require_cuda(probs, "probability loss")
        if probs.dim() != 4:
            raise ValueError(f"expected probabilities [N,C,H,W], got {tuple(probs.shape)}")
        N, C, H, W = probs.shape
        if C > _lib.MAX_CLASSES:
            raise RuntimeError(f"fused loss supports up to {_lib.MAX_CLASSES} classes, got {C}")
        pr = probs.detach()
        if pr.dtype != torch.float32 or not pr.is_contiguous():
            pr = pr.float().contiguous()
        tg = target.detach()
        if tg.dtype != torch.int64 or not tg.is_contiguous():
            tg = tg.long().contiguous()
        if tg.numel() != N * H * W:
            raise ValueError(f"target shape {tuple(target.shape)} does not match probabilities {tuple(probs.shape)}")
        dev = probs.device
        cw = None
        if class_weights is not None:
            cw = class_weights.detach().to(device=dev, dtype=torch.float32).contiguous()
            if cw.numel() != C:
                raise ValueError("class_weights must have one entry per class")
        ign = -1 if ignore_index is None else int(ignore_index)
        part = _f32(_lib.query("segk_loss_part_floats", N * H * W), dev)
        state = _f32(_lib.query("segk_loss_state_floats"), dev)
        out = _f32(1, dev)
        with _span("loss_fwd", 0.0, N * H * W * (4.0 * C + 8)):
            _lib.call("segk_prob_loss_fwd", pr.data_ptr(), tg.data_ptr(), _p(cw), N, C, H * W, ign, float(smooth),
                      float(dice_weight), float(nll_weight), int(nll_log), float(eps), part.data_ptr(), state.data_ptr(),
                      out.data_ptr(), _stream())
        ctx.cfg = (N, C, H, W, ign, float(dice_weight), float(nll_weight), int(nll_log), float(eps))
        ctx.save_for_backward(pr, tg, cw, state)
        return out.view(())

    @staticmethod
    def backward(ctx, gout):
        pr, tg, cw, state = ctx.saved_tensors
        N, C, H, W, ign, dw, nw, nll_log, eps = ctx.cfg
        go = gout.detach().float().reshape(1).contiguous()
        dp = torch.empty_like(pr)
        with _span("loss_bwd", 0.0, N * H * W * (8.0 * C + 8)):
            _lib.call("segk_prob_loss_bwd", pr.data_ptr(), tg.data_ptr(), _p(cw), state.data_ptr(), go.data_ptr(), N, C,
                      H * W, ign, dw, nw, nll_log, eps, dp.data_ptr(), _stream())
        return dp, None, None, None, None, None, None, None, None


class PromptMixFn(torch.autograd.Function):
    """final_probs of the prompt model (reference prompt_based/prompt.py:35-56): softmax of the frozen 4-class CLIP-UNet
    logits remixed with the sigmoid of the mask U-Net's logit; gradient flows to the mask logit only (prompt.py:30-31)."""

    @staticmethod
    def forward(ctx, clip_logits, mask_logit):
        _require_cuda(clip_logits, "prompt remix")
        _require_cuda(mask_logit, "prompt remix")
        N, C, H, W = clip_logits.shape
        if C != 4 or tuple(mask_logit.shape) != (N, 1, H, W):
            raise ValueError(f"prompt remix expects [N,4,H,W] CLIP logits and a [N,1,H,W] mask logit, got "
                             f"{tuple(clip_logits.shape)} and {tuple(mask_logit.shape)}")
        if clip_logits.requires_grad:
            raise RuntimeError("prompt remix: the CLIP branch must be frozen (prompt.py:30-31); only the mask gets a gradient")
        cl = clip_logits.detach().float().contiguous()
        ml = mask_logit.detach().float().contiguous()
        out = torch.empty_like(cl)
        with _span("prompt_mix", 0.0, N * H * W * 36.0):
            _lib.call("segk_prompt_mix_fwd", cl.data_ptr(), ml.data_ptr(), out.data_ptr(), N, H * W, _stream())
        ctx.save_for_backward(cl, ml)
        return out

    @staticmethod
    def backward(ctx, dout):
        cl, ml = ctx.saved_tensors
        N, _, H, W = cl.shape
        d = dout.detach().float().contiguous()
        dm = torch.empty_like(ml)
        with _span("prompt_mix", 0.0, N * H * W * 40.0):
            _lib.call("segk_prompt_mix_bwd", cl.data_ptr(), ml.data_ptr(), d.data_ptr(), dm.data_ptr(), N, H * W, _stream())
        return None, dm


def confusion_matrix(logits, labels, num_classes, out=None):
    """argmax over classes (first maximum, like torch.argmax) + confusion counts on device.
    logits [N,C,H,W] or [C,H,W]; labels [N,H,W] / [H,W].  Returns int64 [num_classes, num_classes] with
    M[pred, label] (reference utils/MetricsHistory.py:65-75 derives TP/FP/FN/TN from exactly these)."""
    _require_cuda(logits, "confusion_matrix")
    if logits.dim() == 3:
        logits = logits.unsqueeze(0)
    N, C, H, W = logits.shape
    if C != num_classes or C > _lib.MAX_CLASSES:
        raise ValueError(f"expected {num_classes} (<= {_lib.MAX_CLASSES}) class channels, got {C}")
    lg = logits.detach()
    if lg.dtype != torch.float32 or not lg.is_contiguous():
        lg = lg.float().contiguous()
    tg = labels.detach()
    if tg.dtype != torch.int64 or not tg.is_contiguous():
        tg = tg.long().contiguous()
    if tg.numel() != N * H * W:
        raise ValueError("label shape does not match logits")
    # `out`: a persistent [MAX_CLASSES, MAX_CLASSES] int64 device matrix the counts are ADDED to (no allocation, no sync)
    M = out if out is not None else torch.zeros((_lib.MAX_CLASSES, _lib.MAX_CLASSES), dtype=torch.int64, device=logits.device)
    _lib.call("segk_confusion", lg.data_ptr(), tg.data_ptr(), N, C, H * W, M.data_ptr(), _stream())
    return M[:C, :C]
