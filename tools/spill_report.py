#!/usr/bin/env python3
"""Register report of every kernel of the library (hipcc -Rpass-analysis=kernel-resource-usage): prints the kernels that
spill or use scratch memory, or all of them with --all.  A spill inside a loop costs more than most tuning gains; one in a
cold tail still makes the launch allocate scratch.  Usage: python tools/spill_report.py [--all] [unit ...]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "image_segmentation_amd", "csrc")


def report(unit):
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "-I" + CSRC,
           "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(CSRC, unit + ".hip"), "-o", "/dev/null"]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in err.splitlines():
        m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|VGPRs Spill|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]):\s+(\S+)", line)
        if not m:
            continue
        key, val = m.group(1), m.group(2)
        if key == "Function Name":
            cur = {"name": val}
            rows.append(cur)
        elif cur is not None:
            cur[key.split(" [")[0]] = val
    return rows


def main():
    show_all = "--all" in sys.argv
    units = [a for a in sys.argv[1:] if not a.startswith("--")] or sorted(f[:-4] for f in os.listdir(CSRC) if f.endswith(".hip"))
    bad = 0
    for u in units:
        for r in report(u):
            spill, scratch = int(r.get("VGPRs Spill", 0)), int(r.get("ScratchSize", 0))
            if show_all or spill or scratch:
                name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()[:110]
                print(f"{u:12s} vgpr {r.get('VGPRs','?'):>4s} agpr {r.get('AGPRs','?'):>4s} spill {spill:4d} scratch {scratch:5d} occ {r.get('Occupancy','?'):>2s}  {name}")
            bad += 1 if (spill or scratch) else 0
    print(f"{bad} kernel(s) with spills or scratch")


if __name__ == "__main__":
    main()
