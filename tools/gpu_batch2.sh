#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "wgrad or biased or convt" > $O/t2_wgrad.log 2>&1; echo "wgrad tests rc=$?"; tail -3 $O/t2_wgrad.log
for i in 1 2; do
timeout -k 10 200 python tools/kbench.py wgrad --lib tools/ubench/bin/libsegk_oldwgrad.so > $O/kb2_wold$i.log 2>&1
timeout -k 10 200 python tools/kbench.py wgrad > $O/kb2_wnew$i.log 2>&1
done
paste $O/kb2_wold1.log $O/kb2_wnew1.log | cut -c1-50,90-170
paste $O/kb2_wold2.log $O/kb2_wnew2.log | cut -c1-50,90-170
timeout -k 10 1500 python -m pytest tests -m gpu -q > $O/gputests2.log 2>&1; echo "tests rc=$?"; tail -8 $O/gputests2.log
timeout -k 10 400 python bench.py > $O/bench2.json 2> $O/bench2.err; echo "bench rc=$?"; tail -3 $O/bench2.err
