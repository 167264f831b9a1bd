#!/bin/bash
# Collects the judged evidence on the MI355X box (run through gpurun from the repo root):
#   bench line (default run), rocprofv3 kernel stats of the same command, FETCH_SIZE / WRITE_SIZE PMC passes.
# Output under gpurun_out/final/ ; copy the summaries into profiles/ afterwards.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o r -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0 > /dev/null 2> $OUT/pmc_fetch.err
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o r -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0 > /dev/null 2> $OUT/pmc_write.err
echo "write done"
python - <<PY
import csv, json, collections, re, sys
sys.path.insert(0, "$R")
from image_segmentation_amd import _lib
out = "$OUT"
def short(name):
    """kernel name without its argument list; template arguments kept (they tell the variants apart)"""
    n = re.sub(r"^void ", "", name)
    n = re.sub(r"\\(anonymous namespace\\)::", "", n)
    d, k = 0, len(n)
    for i, ch in enumerate(n):          # cut at the '(' that opens the argument list (depth 0 outside <...>)
        if ch == "<": d += 1
        elif ch == ">": d -= 1
        elif ch == "(" and d == 0:
            k = i
            break
    return n[:k].strip()
def fam(name):
    if "conv3x3_pipe" in name or "conv_ws_kernel" in name or "conv_rs_kernel" in name or "stem_stream_kernel" in name: return "conv3x3"
    if "convt_stream_kernel" in name: return "conv_other"
    if "conv_igemm_kernel" in name:
        return "conv3x3" if ("Li0ELi" in name.split("conv_igemm_kernel")[1][:12] or "E, 0," in name or ", 0, " in name.split("conv_igemm_kernel")[1][:24]) else "conv_other"
    if "wgrad_dma_kernel" in name: return "wgrad3x3"
    if "wgrad_kernel" in name: return "wgrad_other"
    return None
per = collections.defaultdict(lambda: {"FETCH_SIZE": [], "WRITE_SIZE": []})
agg = collections.defaultdict(lambda: {"FETCH_SIZE": [], "WRITE_SIZE": []})
for ctr, d in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    for r in csv.DictReader(open(f"{out}/{d}/r_counter_collection.csv")):
        if r["Counter_Name"] != ctr: continue
        v = float(r["Counter_Value"])
        per[short(r["Kernel_Name"])][ctr].append(v)
        f = fam(r["Kernel_Name"])
        if f: agg[f][ctr].append(v)
def row(v):
    fe = sum(v["FETCH_SIZE"]) / len(v["FETCH_SIZE"]); wr = sum(v["WRITE_SIZE"]) / len(v["WRITE_SIZE"])
    return {"launches_sampled": len(v["FETCH_SIZE"]), "FETCH_SIZE_KB": round(fe, 1), "WRITE_SIZE_KB": round(wr, 1),
            "hbm_bytes_per_launch": int((2 * fe + wr) * 1024)}
res = {"build_id": _lib.build_id(),
       "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0 (separate passes)",
       "unit": "KB per launch (rocprofv3 raw); hbm_bytes_per_launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: FETCH_SIZE doubled on gfx950 (MI355X_MICROARCH.md HBM section)",
       "families": {f: row(v) for f, v in agg.items() if v["FETCH_SIZE"] and v["WRITE_SIZE"]},
       "kernels": {k: row(v) for k, v in sorted(per.items()) if v["FETCH_SIZE"] and v["WRITE_SIZE"]}}
json.dump(res, open(f"{out}/pmc_traffic.json", "w"), indent=1)
print(json.dumps(res["families"], indent=1))
PY
# the bench attaches a PMC record only when its build_id matches the library: put this run's record where bench.py looks
cp $OUT/pmc_traffic.json $R/profiles/r02_pmc_traffic.json
timeout -k 10 500 python $R/bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o r -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
echo "stats done"

cp $OUT/stats/r_kernel_stats.csv $OUT/kernel_stats.csv
ls -la $OUT
