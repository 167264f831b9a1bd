#!/bin/bash
# rocprofv3 kernel stats of the default bench step -> gpurun_out/r3/stats_<tag>.csv  (usage: tools/rocprof_stats.sh tag)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$1 -o r -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --profile-steps 0 > $O/bench_prof_$1.json 2> $O/prof_$1.err
cp $O/prof_$1/r_kernel_stats.csv $O/stats_$1.csv
python $R/tools/pmc_post.py stats $O/stats_$1.csv 13
