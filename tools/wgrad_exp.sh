#!/bin/bash
# GPU box: what does the filter-gradient kernel wait for?  Diagnostic builds of conv_wgrad.hip with parts of its tile loop
# switched off (XV_WGRAD_EXP bits: 1 no LDS-DMA after the first tile, 2 no per-tile barrier, 4 fragments of the first row
# only; wrong results, same instruction stream otherwise) against the real kernel, one box, one call.
#   build first (CPU):  bash tools/wgrad_exp.sh build      then:  gpurun -- 'bash tools/wgrad_exp.sh run r6'
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
CS=$ROOT/modular_semantic_segmentation_amd/csrc
B=$ROOT/tools/build/wgrad_exp
if [ "$1" = build ]; then
  mkdir -p $B
  make -C $CS -j8 > /dev/null || exit 1
  for e in ${2:-1 2 3 4 7}; do
    hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -DXV_WGRAD_EXP=$e -c $CS/conv_wgrad.hip -o $B/conv_wgrad_$e.o || exit 1
    objs=$(ls $CS/build/*.o | grep -v conv_wgrad.o)
    hipcc --offload-arch=gfx950 -shared -fPIC $objs $B/conv_wgrad_$e.o -o $B/libxview_hip_wgrad_exp$e.so || exit 1
  done
  exit 0
fi
TAG=${2:-r6}
OUT=$ROOT/gpurun_out/${TAG}_wgrad_exp.txt
cd $ROOT
: > $OUT
L=${3:-conv2_2,conv3_2,conv4_2,conv5_1}
for rep in 1 2; do
  echo "== real kernel (pass $rep)" >> $OUT; python3 tools/wgrad_bench.py --layers $L >> $OUT 2>&1
  for f in $B/libxview_hip_wgrad_exp*.so; do
    echo "== $(basename $f) (pass $rep)" >> $OUT
    XV_ALLOW_STALE_LIB=1 XV_LIB=$f python3 tools/wgrad_bench.py --layers $L >> $OUT 2>&1
  done
done
cat $OUT
