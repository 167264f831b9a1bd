#!/bin/bash
# A/B library variants for same-box comparisons (never shipped): tools/ubench/bin/libsegk_<NAME>.so, the current objects with
# some translation units replaced.  Each further argument is  unit[:source][:flags]  -- unit = csrc file stem, source = an
# alternative .hip file (default: the unit's own source), flags = extra compiler flags (quote them).
#   tools/variant_build.sh oldwgrad wgrad:/tmp/wgrad_old.hip
#   tools/variant_build.sh ring4 conv_rs::"-DRS_RING=4"
# then  python tools/kbench.py wgrad --lib tools/ubench/bin/libsegk_oldwgrad.so   (same ABI as the shipped library: the
# loader refuses anything else)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/image_segmentation_amd/csrc
NAME=$1; shift
D=$R/tools/ubench/bin/var_$NAME
mkdir -p $D
python -c "import sys; sys.path.insert(0, '$R'); from image_segmentation_amd import build; build.build(verbose=False)"
cp $C/*.o $D/
for spec in "$@"; do
  IFS=':' read -r unit src flags <<< "$spec"
  [ -z "$src" ] && src=$C/$unit.hip
  extra=""
  [ "$unit" = api ] && extra="-DSEGK_BUILD_ID=\"$(cat $C/.build_id)\""
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -I$C $flags $extra -c $src -o $D/$unit.o
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/ubench/bin/libsegk_$NAME.so $D/*.o
rm -rf $D
echo built $R/tools/ubench/bin/libsegk_$NAME.so
