#!/bin/bash
# Diagnostic library with in-kernel phase stamps (never shipped): tools/ubench/bin/libsegk_stamp.so
# usage: tools/stamp_build.sh   then   python tools/kbench.py conv --only 64->64 --lib tools/ubench/bin/libsegk_stamp.so --stamps
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/image_segmentation_amd/csrc
mkdir -p $R/tools/ubench/bin/stamp_obj
for f in api conv_igemm conv_rs convt_stream stem wgrad bn_pool pack head_loss resize vit gemm probe; do
  if [ "$f" = conv_rs ]; then X="-DSEGK_RS_STAMPS ${RS_ABL:+-DRS_ABL=$RS_ABL} ${RS_EXTRA}";
  elif [ "$f" = conv_igemm ]; then X="-DSEGK_PIPE_STAMPS ${PIPE_EXTRA}";
  elif [ "$f" = wgrad ]; then X="-DSEGK_WGRAD_STAMPS ${WGRAD_EXTRA}";
  elif [ "$f" = bn_pool ]; then X="${POOL_EXTRA}"; else X=""; fi
  if [ "$f" = conv_rs ] || [ "$f" = conv_igemm ] || [ "$f" = wgrad ] || [ "$f" = bn_pool ] || [ ! -f $C/$f.o ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off $X -c $C/$f.hip -o $R/tools/ubench/bin/stamp_obj/$f.o
  else
    cp $C/$f.o $R/tools/ubench/bin/stamp_obj/$f.o
  fi
done
OUT=$R/tools/ubench/bin/libsegk_stamp${RS_ABL:+_abl$RS_ABL}${RS_TAG}.so   # RS_TAG also names PIPE_EXTRA builds
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $R/tools/ubench/bin/stamp_obj/*.o
python3 -c "import ctypes,sys; ctypes.CDLL(sys.argv[1])" $OUT   # every symbol resolves (a missing unit would only show on the GPU box)
echo built $OUT
