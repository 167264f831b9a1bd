#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3
mkdir -p $O; cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "k_split or persistent_units or side_output" > $O/t4_rk.log 2>&1; rc=$?; echo "rk test rc=$rc"; tail -5 $O/t4_rk.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 100 env SEGK_NO_RK=1 python tools/kbench.py conv --only "128-" --no-stats > $O/kb4_pipe.log 2>&1; grep conv $O/kb4_pipe.log
timeout -k 10 100 python tools/kbench.py conv --only "128-" --no-stats > $O/kb4_base.log 2>&1; grep conv $O/kb4_base.log
for v in 11 9 3; do
  timeout -k 10 100 python tools/kbench.py conv --only "128->128" --no-stats --lib tools/ubench/bin/libsegk_rkabl$v.so > $O/kb4_abl$v.log 2>&1; echo "abl $v: $(grep conv $O/kb4_abl$v.log)"
done
