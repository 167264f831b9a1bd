#!/bin/bash
# rocprofv3 kernel stats of another bench configuration -> gpurun_out/r3/stats_<tag>.csv   (usage: tools/rocprof_cfg.sh tag bench-args...)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3
TAG=$1; shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$TAG -o r -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --profile-steps 0 "$@" > $O/bench_prof_$TAG.json 2> $O/prof_$TAG.err
cp $O/prof_$TAG/r_kernel_stats.csv $O/stats_$TAG.csv
python $R/tools/pmc_post.py stats $O/stats_$TAG.csv 13
