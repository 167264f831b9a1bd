"""Build libsegk.so (hand-written HIP kernels for gfx950) in-tree with hipcc.

No torch dependency: the library is a plain C-ABI shared object (include/segk.h).  hipcc cross-compiles
gfx950 without a GPU present, so this also runs in the CPU-only build container.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
SOURCES = ["api.hip", "conv_igemm.hip", "conv_rs.hip", "convt_stream.hip", "stem.hip", "wgrad.hip", "bn_pool.hip", "pack.hip", "head_loss.hip", "resize.hip", "vit.hip", "gemm.hip", "probe.hip"]
LIB = os.path.join(CSRC, "libsegk.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def source_hash() -> str:
    """sha256 over every file the library is built from (kernel sources, internal headers, the public header), in a
    fixed order.  Compiled into the library (segk_build_id()), so a shipped libsegk.so can be checked against the
    sources beside it (tests/test_abi.py) and profile records can name the build they were taken on."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp", ".h")))
    paths = [os.path.join(CSRC, f) for f in files] + \
            [os.path.join(os.path.dirname(os.path.dirname(CSRC)), "include", "segk.h")]
    for path in paths:
        h.update(os.path.basename(path).encode() + b"\0")
        h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


def _stale(obj, src):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    deps = [src] + [os.path.join(CSRC, h) for h in ("common.hpp", "segk_internal.h")] + \
           [os.path.join(os.path.dirname(os.path.dirname(CSRC)), "include", "segk.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = _hipcc()
    objs, jobs = [], []
    bid = source_hash()
    stamp = os.path.join(CSRC, ".build_id")
    id_changed = not os.path.exists(stamp) or open(stamp).read().strip() != bid
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(CSRC, s.replace(".hip", ".o"))
        objs.append(obj)
        extra = [f'-DSEGK_BUILD_ID="{bid}"'] if s == "api.hip" else []     # api.hip carries the id: rebuilt when it moves
        if force or _stale(obj, src) or (extra and id_changed):
            jobs.append([hipcc, *FLAGS, *extra, "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=6) as ex:
        list(ex.map(run, jobs))
    if jobs or not os.path.exists(LIB):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
    with open(stamp, "w") as f:
        f.write(bid + "\n")
    return LIB


def build_sanitized(verbose: bool = False) -> str:
    """Host-side AddressSanitizer + UndefinedBehaviorSanitizer build of the library (device code is compiled as usual:
    GPU sanitizers are not available on this pool) -> csrc/.asan/libsegk_asan.so, rebuilt when the source hash moves.
    Used by tests/test_abi_hardening.py, which drives every C-ABI entry with invalid arguments on the CPU box."""
    hipcc = _hipcc()
    out = os.path.join(CSRC, ".asan")
    os.makedirs(out, exist_ok=True)
    lib = os.path.join(out, "libsegk_asan.so")
    stamp = os.path.join(out, ".build_id")
    bid = source_hash()
    if os.path.exists(lib) and os.path.exists(stamp) and open(stamp).read().strip() == bid:
        return lib
    flags = ["--offload-arch=gfx950", "-O1", "-g", "-fPIC", "-std=c++17", "-ffp-contract=off", "-fno-omit-frame-pointer",
             "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-gpu-sanitize"]
    objs, jobs = [], []
    for s in SOURCES:
        obj = os.path.join(out, s.replace(".hip", ".o"))
        objs.append(obj)
        extra = [f'-DSEGK_BUILD_ID="{bid}"'] if s == "api.hip" else []
        jobs.append([hipcc, *flags, *extra, "-c", os.path.join(CSRC, s), "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
    with ThreadPoolExecutor(max_workers=6) as ex:
        list(ex.map(run, jobs))
    run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fsanitize=address,undefined", "-shared-libsan", "-o", lib, *objs])
    with open(stamp, "w") as f:
        f.write(bid + "\n")
    return lib


def asan_runtime() -> str:
    """The shared AddressSanitizer runtime to LD_PRELOAD into an uninstrumented python."""
    import glob
    hits = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    if not hits:
        raise RuntimeError("libclang_rt.asan-x86_64.so not found under /opt/rocm/lib/llvm")
    return hits[-1]


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
