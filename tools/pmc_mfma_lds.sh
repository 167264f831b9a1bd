#!/bin/bash
# MFMA busy cycles, LDS bank-conflict cycles and LDS active cycles per kernel of the bench step (separate PMC passes, no
# tracing domains besides --kernel-trace): gpurun_out/final/pmc_mfma_lds.json.  Run through gpurun from the repo root.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -o r -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0 --no-extras > /dev/null 2> $OUT/pmc_$C.err
  echo "$C done"
done
python $R/tools/pmc_post.py mfma_lds $OUT $R
