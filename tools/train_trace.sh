#!/bin/bash
# Run on the GPU box: rocprofv3 kernel stats of the training step with everything on ONE stream (XV_WGRAD_STREAM=0: a
# kernel's duration is then its own -- beside the data-gradient convs of the other stream the small kernels wait for
# registers and their recorded durations grow tenfold), with and without the routed pool.
TAG=${1:-r5}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
export XV_WGRAD_STREAM=0
TRAIN="python3 $ROOT/bench.py --mode train --steps 3 --warmup 1 --min-seconds 0 --no-cpu-baseline --no-accuracy --no-extra"
run() { d=$1; shift; rm -rf $OUT/$d; rocprofv3 --output-format csv "$@" > $OUT/$d.log 2>&1; }
run ${TAG}_trace_train_serial --kernel-trace --stats -d $OUT/${TAG}_trace_train_serial -o bench -- $TRAIN
export XV_ROUTED_POOL=0
run ${TAG}_trace_train_serial_unrouted --kernel-trace --stats -d $OUT/${TAG}_trace_train_serial_unrouted -o bench -- $TRAIN
for d in serial serial_unrouted; do
  f=$(ls $OUT/${TAG}_trace_train_$d/bench_kernel_stats.csv $OUT/${TAG}_trace_train_$d/*/bench_kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && cp $f $OUT/${TAG}_train_${d}_kernel_stats.csv
done
