#!/bin/bash
# Per-layer HBM traffic of the 3x3 conv and weight-gradient kernels at the U-Net layer shapes (B = 32): FETCH_SIZE and WRITE_SIZE
# passes of `tools/kbench.py conv` without and with the BatchNorm+ReLU prologue, and of `tools/kbench.py wgrad` (counters only with --kernel-trace, one counter per pass).
#   tools/pmc_layers.sh <out_dir_under_gpurun_out>      -> gpurun_out/<out>/layers.txt   (tools/pmc_post.py layers)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$1
IT=2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for tag in ${TAGS:-plain pro wgrad}; do
  X=""; [ $tag = pro ] && X="--pro"
  W=conv; [ $tag = wgrad ] && W=wgrad
  for ctr in FETCH_SIZE WRITE_SIZE; do
    d=${tag}_$(echo $ctr | cut -d_ -f1 | tr A-Z a-z)
    timeout -k 10 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/$d -o r -- python $R/tools/kbench.py $W --iters $IT $X --list $OUT/${tag}_layers.txt > $OUT/$d.log 2>&1 || echo "pass $d failed" >> $OUT/errors.txt
  done
done
python $R/tools/pmc_post.py layers $OUT $((IT + 3))
