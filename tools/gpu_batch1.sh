#!/bin/bash
# round 3, batch 1: GPU test suite, default bench, launch census, per-layer kernel table
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3
mkdir -p $O
cd $R
timeout -k 10 1500 python -m pytest tests -m gpu -x -q > $O/gputests1.log 2>&1; echo "tests rc=$?" | tee -a $O/gputests1.log
tail -5 $O/gputests1.log
timeout -k 10 400 python bench.py > $O/bench1.json 2> $O/bench1.err; echo "bench rc=$?"
timeout -k 10 200 python tools/launch_census.py > $O/census1.log 2>&1; echo "census rc=$?"
timeout -k 10 300 python tools/kbench.py all > $O/kb1.log 2>&1; echo "kbench rc=$?"
timeout -k 10 200 python tools/kbench.py conv --pro > $O/kb1_pro.log 2>&1
timeout -k 10 200 python tools/kbench.py convt > $O/kb1_convt.log 2>&1
timeout -k 10 200 python tools/kbench.py bn > $O/kb1_bn.log 2>&1
