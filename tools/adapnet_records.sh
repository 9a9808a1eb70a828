#!/bin/bash
# The AdapNet rows on ONE box: inference record, training record, kernel stats of the serialised step.
#   gpurun --timeout 900 -- tools/adapnet_records.sh r6
TAG=${1:-r6}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
python3 $ROOT/bench.py --expert adapnet --steps 10 --warmup 3 --no-cpu-baseline --no-accuracy --no-extra 2>/dev/null | tail -1 > $OUT/${TAG}_bench_adapnet.json
python3 $ROOT/bench.py --mode train --expert adapnet --batch 8 --steps 5 --warmup 2 --no-cpu-baseline --no-accuracy --no-extra 2>/dev/null | tail -1 > $OUT/${TAG}_bench_adapnet_train.json
AD="python3 $ROOT/bench.py --expert adapnet --steps 3 --warmup 1 --min-seconds 0 --serial-experts --no-graph --no-cpu-baseline --no-accuracy --no-extra"
rm -rf $OUT/${TAG}_trace_adapnet
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/${TAG}_trace_adapnet -o bench -- $AD > $OUT/${TAG}_trace_adapnet.log 2>&1
f=$(ls $OUT/${TAG}_trace_adapnet/bench_kernel_stats.csv $OUT/${TAG}_trace_adapnet/*/bench_kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp $f $OUT/${TAG}_adapnet_kernel_stats.csv
rm -f $OUT/${TAG}_trace_adapnet/bench_kernel_trace.csv $OUT/${TAG}_trace_adapnet/*/bench_kernel_trace.csv
for f in adapnet adapnet_train; do echo "== $f"; cut -c1-400 $OUT/${TAG}_bench_$f.json; done
head -8 $OUT/${TAG}_adapnet_kernel_stats.csv | cut -c1-160
