#!/bin/bash
# Multi-rank rehearsals on the ONE-GPU box (RCCL needs one GPU per rank: these runs share cuda:0 over gloo and prove the launch and
# multi-rank code path only -- the numbers mean nothing): the self-launching bench, the bench under torch.distributed.run as the
# driver starts it, and the 2-rank == single-process gradient check.  Output under gpurun_out/ddp/ ; copy into profiles/<tag>_*.
#   tools/collect_ddp.sh [tag, default r04]
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/ddp
mkdir -p $OUT
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 400 python bench.py --gpus 2 --steps 6 --warmup 2 --backend gloo --share-device --no-cpu-baseline --no-extras \
  > $OUT/${TAG}_2rank_selflaunch_gloo_bench.json 2> $OUT/selflaunch.err
echo "selflaunch rc=$?"
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
  bench.py --gpus 2 --steps 6 --warmup 2 --backend gloo --share-device --no-cpu-baseline --no-extras \
  > $OUT/${TAG}_2rank_torchrun_gloo_bench.json 2> $OUT/torchrun.err
echo "torchrun rc=$?"
timeout -k 10 400 python tools/ddp_check.py $OUT/${TAG}_ddp_check_2rank_gloo.json
echo "ddp_check rc=$?"
ls -la $OUT
