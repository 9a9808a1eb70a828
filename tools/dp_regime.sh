#!/bin/bash
# GPU box: the data-parallel regime on ONE GPU -- the training step at 4 images (the reference's shipped batchsize,
# example_config.yaml:18, and the per-rank batch of a 32-image DP-8 step), plain and batch_normalization=true: bench records and a
# kernel trace each, from which tools/gpu_busy.py reads the GPU-busy fraction (does the chip wait for the host?).
#   gpurun -- 'bash tools/dp_regime.sh r6 [extra bench flags]'
TAG=${1:-r6}; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
TRAIN="python3 $ROOT/bench.py --mode train --no-cpu-baseline --no-accuracy --no-extra"
for b in 4 16; do for v in plain bn; do
  x=""; [ $v = bn ] && x="--batch-norm"
  $TRAIN --batch $b $x --steps 10 --warmup 3 --min-seconds 1.5 "$@" > $OUT/${TAG}_train_b${b}_${v}.json 2> $OUT/${TAG}_train_b${b}_${v}.err
done; done
for v in plain bn; do
  x=""; [ $v = bn ] && x="--batch-norm"
  d=$OUT/${TAG}_trace_train_b4_$v
  rm -rf $d
  rocprofv3 --output-format csv --kernel-trace --stats -d $d -o bench -- $TRAIN --batch 4 $x --steps 40 --warmup 3 --min-seconds 0 --no-roofline-pass "$@" > $d.log 2>&1
  f=$(ls $d/bench_kernel_stats.csv $d/*/bench_kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && cp $f $OUT/${TAG}_train_b4_${v}_kernel_stats.csv
  t=$(ls $d/bench_kernel_trace.csv $d/*/bench_kernel_trace.csv 2>/dev/null | head -1)
  [ -n "$t" ] && python3 $ROOT/tools/gpu_busy.py $t --last 30 > $OUT/${TAG}_train_b4_${v}_gpu_busy.json
  rm -f $t   # (tens of MB)
done
cd $OUT
for f in ${TAG}_train_b*_*.json; do echo "== $f"; cut -c1-420 $f; done
