#!/bin/bash
# GPU box: the records and traces refreshed after a late kernel change (bench records, elementwise table, training traces, the
# Dirichlet A/B): gpurun -- 'bash tools/round_final.sh r5'
TAG=${1:-r5}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
bash $ROOT/tools/bench_records.sh $TAG > $OUT/${TAG}_records.log 2>&1
bash $ROOT/tools/profile_elementwise.sh $TAG > $OUT/${TAG}_elementwise.log 2>&1
bash $ROOT/tools/train_trace.sh $TAG > $OUT/${TAG}_train_trace.log 2>&1
cd /tmp && export TMPDIR=/tmp
unset XV_WGRAD_STREAM XV_ROUTED_POOL
TRAIN="python3 $ROOT/bench.py --mode train --steps 3 --warmup 1 --min-seconds 0 --no-cpu-baseline --no-accuracy --no-extra"
for v in train train_bn; do
  extra=""; [ $v = train_bn ] && extra="--batch-norm"
  rm -rf $OUT/${TAG}_trace_$v
  rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/${TAG}_trace_$v -o bench -- $TRAIN $extra > $OUT/${TAG}_trace_$v.log 2>&1
  f=$(ls $OUT/${TAG}_trace_$v/bench_kernel_stats.csv $OUT/${TAG}_trace_$v/*/bench_kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && cp $f $OUT/${TAG}_${v}_kernel_stats.csv
done
BENCH="python3 $ROOT/bench.py --steps 3 --warmup 1 --min-seconds 0 --serial-experts --no-graph --no-cpu-baseline --no-accuracy --no-extra"
rm -rf $OUT/${TAG}_trace
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/${TAG}_trace -o bench -- $BENCH > $OUT/${TAG}_trace.log 2>&1
f=$(ls $OUT/${TAG}_trace/bench_kernel_stats.csv $OUT/${TAG}_trace/*/bench_kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp $f $OUT/${TAG}_bench_serial_kernel_stats.csv
cd $ROOT
python3 tools/dirichlet_head_ab.py > $OUT/${TAG}_dirichlet_ab.json 2>/dev/null
tail -12 $OUT/${TAG}_records.log | cut -c1-230
tail -24 $OUT/${TAG}_elementwise.log
