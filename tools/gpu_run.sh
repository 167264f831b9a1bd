#!/bin/bash
# Builder convenience: run a list of commands on the GPU box with outputs under gpurun_out/r4/ (created on the box).
#   tools/gpu_run.sh 'cmd1' 'cmd2' ...     each command's stdout/stderr -> gpurun_out/r4/<n>.log, tails echoed
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4
mkdir -p $O; cd $R
i=0
for c in "$@"; do
  i=$((i+1))
  echo "== [$i] $c"
  timeout -k 10 900 bash -c "$c" > $O/run_$i.log 2>&1; echo "rc=$?"
  tail -${TAILN:-12} $O/run_$i.log
done
