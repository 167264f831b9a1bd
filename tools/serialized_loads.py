#!/usr/bin/env python3
"""Static check of the compiled kernels for loads that wait for themselves: a global load followed within three instructions
by `s_waitcnt vmcnt(0)`.  hipcc compiles `cond ? p[i] : 0` (and loads under a divergent `if`) to branch + load + wait, so N
"independent" conditional loads run as N serial memory round trips (found in round 3: every finalize walk of the library).
Usage: python tools/serialized_loads.py [unit ...]     (default: all units)"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "image_segmentation_amd", "csrc")


def scan(unit):
    out = f"/tmp/segk_{unit}.s"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "-I" + CSRC,
                    "--cuda-device-only", "-S", os.path.join(CSRC, unit + ".hip"), "-o", out], check=True, capture_output=True)
    txt = open(out).read()
    rows = []
    for m in re.finditer(r"^(\w+):\s*; @\w+\n(.*?)s_endpgm", txt, re.S | re.M):
        body = m.group(2).splitlines()
        loads = [i for i, l in enumerate(body) if re.search(r"\b(global_load|buffer_load)_\w+", l) and " lds" not in l]
        ser = 0
        for i in loads:
            for j in range(i + 1, min(i + 4, len(body))):
                if "global_load" in body[j] or "buffer_load" in body[j]:
                    break
                if "s_waitcnt vmcnt(0)" in body[j]:
                    ser += 1
                    break
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()[:100]
        rows.append((ser, len(loads), unit, name))
    return rows


def main():
    units = sys.argv[1:] or sorted(f[:-4] for f in os.listdir(CSRC) if f.endswith(".hip"))
    rows = [r for u in units for r in scan(u)]
    for ser, n, u, name in sorted(rows, reverse=True):
        if ser >= 2:
            print(f"{u:12s} {ser:3d} of {n:3d} loads wait for themselves   {name}")


if __name__ == "__main__":
    main()
