#!/bin/bash
# HBM write bytes per conv layer for two library variants (same-box): tools/ubench/pmc_write_ab.sh <variantA> <variantB> [kbench --only filter]
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r4/write_ab
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in $1 $2; do
  timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/$v -o r -- python $R/tools/kbench.py conv --iters 2 ${3:+--only $3} --lib $R/tools/ubench/bin/libsegk_$v.so > $OUT/$v.log 2>&1
  python $R/tools/ubench/pmc_write_print.py $OUT/$v/r_counter_collection.csv $v
done
