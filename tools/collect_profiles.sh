#!/bin/bash
# Collects the judged evidence on the MI355X box (run through gpurun from the repo root):
#   bench line (default run), rocprofv3 kernel stats of the same command, FETCH_SIZE / WRITE_SIZE PMC passes.
# Output under gpurun_out/final/ ; copy the summaries into profiles/ afterwards.
#   tools/collect_profiles.sh [round tag, default r03]
set -e
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o r -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0 --no-extras > /dev/null 2> $OUT/pmc_fetch.err
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o r -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0 --no-extras > /dev/null 2> $OUT/pmc_write.err
echo "write done"
python $R/tools/pmc_post.py traffic $OUT $R
# the bench attaches a PMC record only when its build_id matches the library: put this run's record where bench.py looks
cp $OUT/pmc_traffic.json $R/profiles/${TAG}_pmc_traffic.json
timeout -k 10 500 python $R/bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o r -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
echo "stats done"

cp $OUT/stats/r_kernel_stats.csv $OUT/kernel_stats.csv
ls -la $OUT
