#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3
mkdir -p $O
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "k_split" > $O/t3_rk.log 2>&1; rc=$?; echo "rk test rc=$rc"; tail -15 $O/t3_rk.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q > $O/t3_kernels.log 2>&1; echo "kernel tests rc=$?"; tail -3 $O/t3_kernels.log
timeout -k 10 300 env SEGK_NO_RK=1 python tools/kbench.py conv --only 128- > $O/kb3_nork.log 2>&1
timeout -k 10 300 python tools/kbench.py conv --only 128- > $O/kb3_rk.log 2>&1
paste $O/kb3_nork.log $O/kb3_rk.log | cut -c1-60,95-175
timeout -k 10 300 env SEGK_NO_RK=1 python tools/kbench.py conv --only 128- --pro > $O/kb3_nork_pro.log 2>&1
timeout -k 10 300 python tools/kbench.py conv --only 128- --pro > $O/kb3_rk_pro.log 2>&1
paste $O/kb3_nork_pro.log $O/kb3_rk_pro.log | cut -c1-60,95-175
timeout -k 10 300 python tools/kbench.py conv > $O/kb3_all.log 2>&1
