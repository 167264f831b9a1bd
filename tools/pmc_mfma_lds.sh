#!/bin/bash
# MFMA busy cycles, LDS bank-conflict cycles and LDS active cycles per kernel of the bench step (separate PMC passes, no
# tracing domains besides --kernel-trace): gpurun_out/final/pmc_mfma_lds.json.  Run through gpurun from the repo root.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -o r -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0 > /dev/null 2> $OUT/pmc_$C.err
  echo "$C done"
done
python - <<PY
import csv, json, collections, re, sys
sys.path.insert(0, "$R")
from image_segmentation_amd import _lib
out = "$OUT"
def short(name):
    n = re.sub(r"^void ", "", name); n = re.sub(r"\\(anonymous namespace\\)::", "", n)
    d, k = 0, len(n)
    for i, ch in enumerate(n):
        if ch == "<": d += 1
        elif ch == ">": d -= 1
        elif ch == "(" and d == 0:
            k = i; break
    return n[:k].strip()
ctrs = ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "GRBM_GUI_ACTIVE"]
per = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ctrs:
    for r in csv.DictReader(open(f"{out}/pmc_{c}/r_counter_collection.csv")):
        if r["Counter_Name"] == c:
            per[short(r["Kernel_Name"])][c].append(float(r["Counter_Value"]))
res = {"build_id": _lib.build_id(),
       "command": "rocprofv3 --pmc <one counter> --kernel-trace -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0 (one pass per counter)",
       "note": "averages per launch, summed over the chip as rocprofv3 reports them; GRBM_GUI_ACTIVE comes summed over the 8 XCDs (16.8 "GHz" against wall time), so mfma_duty = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs) is the matrix-pipe duty cycle; lds_conflict_share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE",
       "kernels": {}}
for k, v in sorted(per.items()):
    if not all(c in v and v[c] for c in ctrs): continue
    a = {c: sum(v[c]) / len(v[c]) for c in ctrs}
    if a["GRBM_GUI_ACTIVE"] < 20000: continue          # tiny kernels
    res["kernels"][k] = {"launches_sampled": len(v[ctrs[0]]), **{c: round(a[c], 1) for c in ctrs},
                         "mfma_duty": round(a["SQ_VALU_MFMA_BUSY_CYCLES"] / (a["GRBM_GUI_ACTIVE"] / 8 * 1024), 4),
                         "lds_conflict_share": round(a["SQ_LDS_BANK_CONFLICT"] / a["SQ_LDS_IDX_ACTIVE"], 4) if a["SQ_LDS_IDX_ACTIVE"] else None}
json.dump(res, open(f"{out}/pmc_mfma_lds.json", "w"), indent=1)
for k, v in res["kernels"].items():
    print(f"{k[:60]:60s} mfma {v['mfma_duty']:.3f}  lds_conflict {v['lds_conflict_share']}")
PY
