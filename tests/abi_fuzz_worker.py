"""Child process of tests/test_abi_hardening.py: loads a (sanitizer-instrumented) build of libsegk.so and calls EVERY
int-returning entry of the C ABI with invalid arguments -- NULL pointers, zero / negative / misaligned sizes, bad enum
values.  Contract under test (include/segk.h): every such call returns a negative code from ARGUMENT VALIDATION (-2:
nothing was launched, this box has no GPU to launch on) and leaves a message in segk_last_error(); the pure size queries
never crash.  Prints one JSON object."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from image_segmentation_amd import _lib          # noqa: E402  (signature table only; the library path comes from argv)

lib = C.CDLL(sys.argv[1])
lib.segk_last_error.restype = C.c_char_p
QUERIES = {"segk_version", "segk_entry_count", "segk_stem3x3_wgrad_slabs", "segk_stem3x3_rows", "segk_pack_convt_chunk", "segk_conv_tiles", "segk_bn_stats_floats", "segk_conv_writes_act_q", "segk_wgrad_tiles", "segk_wgrad_split",
           "segk_bn_bwd_blocks", "segk_maxpool_bwd_stat_blocks", "segk_head_part_floats", "segk_head_bwd_blocks",
           "segk_loss_part_floats", "segk_loss_state_floats"}
SPECIAL = {"segk_wgrad_reduce_multi"}           # takes a host array of structs: driven by its own patterns below


def args_for(argtypes, ints, ptr):
    out = []
    k = 0
    for t in argtypes:
        if t is C.c_void_p:
            out.append(C.c_void_p(ptr))
        elif t in (C.c_float, C.c_double):
            out.append(t(1.0))
        else:
            out.append(t(ints[k % len(ints)]))
            k += 1
    return out


def dtype_positions():
    """function name -> index of its `int dtype` parameter, read from include/segk.h"""
    import re
    txt = open(os.path.join(ROOT, "include", "segk.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    pos = {}
    for m in re.finditer(r"\bint\s+(segk_\w+)\s*\(([^)]*)\)\s*;", txt):
        params = [p.strip() for p in m.group(2).split(",")]
        for i, prm in enumerate(params):
            if re.search(r"\bint\s+dtype$", prm):
                pos[m.group(1)] = i
    return pos


DTYPE_AT = dtype_positions()
bad = []
calls = 0
for name, (res, argtypes) in sorted(_lib.SIGNATURES.items()):
    fn = getattr(lib, name)
    fn.restype = res
    fn.argtypes = argtypes
    if res is C.c_char_p:
        fn()
        continue
    if name in SPECIAL:                          # host-read job array: NULL, a bad count, and jobs with NULL / bad fields
        jobs = (_lib.ReduceJob * 4)()
        cases = [(None, 1), (jobs, 0), (jobs, 5), (jobs, 2)]
        jobs2 = (_lib.ReduceJob * 1)(_lib.ReduceJob(256, 256, 7, 1, 1, 1, 0, 32, 32, 0, 1, 0))      # unknown kind
        jobs3 = (_lib.ReduceJob * 1)(_lib.ReduceJob(256, 256, 0, 0, -1, 33, 0, 32, 32, 0, 9, 0))    # nonsense sizes
        jobs4 = (_lib.ReduceJob * 1)(_lib.ReduceJob(256, 256, 1, 4, 64, 60, 8, 0, 0, 0, 0, 0))      # column range past the row
        cases += [(jobs2, 1), (jobs3, 1), (jobs4, 1)]
        for arr, n in cases:
            print(f"call {name} n={n}", file=sys.stderr, flush=True)
            rc = fn(arr, n, None)
            calls += 1
            msg = lib.segk_last_error() or b""
            if rc != -2 or not msg:
                bad.append({"fn": name, "ints": [n], "ptr": 0, "rc": rc, "msg": msg.decode(errors="replace")[:120]})
        continue
    if name in QUERIES:
        for ints in ([0], [-1], [1], [7, 3], [1 << 20], [2, 32, 32, 64, 64, 1], [1, 1, 1, 1, 1, 0]):
            print(f"call {name} {ints}", file=sys.stderr, flush=True)     # a crash is attributed by the last line
            fn(*args_for(argtypes, ints, 0))     # must not crash; any value is acceptable for nonsense input
            calls += 1
        continue
    patterns = [([0], 0),                        # all sizes zero, all pointers NULL
                ([-1], 0),                       # negative sizes
                ([2, 8, 8, 32, 32, 1], 0),       # plausible sizes, NULL pointers
                ([2, 8, 8, 33, 31, 1], 256),     # non-NULL (never dereferenced by the host) pointers, channel counts not multiples of 32
                ([1, 4, 4, 32, 32, 9], 256)]     # bad dtype / enum values
    for k, (ints, ptr) in enumerate(patterns):
        print(f"call {name} {ints} ptr={ptr}", file=sys.stderr, flush=True)
        rc = fn(*args_for(argtypes, ints, ptr))
        calls += 1
        msg = lib.segk_last_error() or b""
        # NULL pointers / zero / negative sizes are always refused by validation (-2).  With non-NULL pointers and sizes
        # that happen to be plausible for this entry the call may be legitimate: it then fails at the launch (-3: this box
        # has no GPU) -- never succeeds, never crashes.
        ok = (rc == -2) if k < 3 else (rc in (-2, -3))
        if not ok or not msg:
            bad.append({"fn": name, "ints": ints, "ptr": ptr, "rc": rc, "msg": msg.decode(errors="replace")[:120]})
    if name in DTYPE_AT:                         # otherwise well-formed call with an unknown dtype: refused before any launch
        a = args_for(argtypes, [1, 4, 4, 32, 32, 32, 32, 1, 1, 1], 256)
        a[DTYPE_AT[name]] = C.c_int(9)
        print(f"call {name} dtype=9", file=sys.stderr, flush=True)
        rc = fn(*a)
        calls += 1
        if rc != -2:
            bad.append({"fn": name, "ints": "dtype=9", "ptr": 256, "rc": rc,
                        "msg": (lib.segk_last_error() or b"").decode(errors="replace")[:120]})
print(json.dumps({"calls": calls, "bad": bad}))
