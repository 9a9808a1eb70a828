#!/bin/bash
# GPU box: round 6's records and traces in one call (gpurun --timeout 2400 -- 'bash tools/round6_records.sh r6'):
# the bench records of tools/bench_records.sh, the DP-regime records with their kernel traces / GPU-busy fractions
# (tools/dp_regime.sh), configs[3]'s fit record, AdapNet re-measured, the training traces (one stream), the serialised
# inference trace the roofline is checked against, the filter-gradient counters.
TAG=${1:-r6}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
bash $ROOT/tools/bench_records.sh $TAG > $OUT/${TAG}_records.log 2>&1
cd $ROOT
python3 bench.py --record dirichlet_fit > $OUT/${TAG}_bench_dirichlet_fit_2048.json 2>/dev/null
python3 bench.py --record dp_regime > $OUT/${TAG}_bench_dp_regime.json 2>/dev/null
python3 bench.py --expert adapnet --steps 10 --warmup 3 --no-cpu-baseline --no-accuracy --no-extra > $OUT/${TAG}_bench_adapnet.json 2>/dev/null
python3 bench.py --mode train --expert adapnet --batch 8 --steps 5 --warmup 2 --no-cpu-baseline --no-accuracy --no-extra > $OUT/${TAG}_bench_adapnet_train.json 2>/dev/null
python3 bench.py --fusion joint --steps 10 --warmup 3 --no-cpu-baseline --no-accuracy --no-extra > $OUT/${TAG}_bench_fusion_fcn.json 2>/dev/null
python3 tools/wgrad_bench.py --variant 2 > $OUT/${TAG}_wgrad_layers_v2.txt 2>/dev/null
python3 tools/wgrad_bench.py --variant 3 > $OUT/${TAG}_wgrad_layers_v3.txt 2>/dev/null
bash $ROOT/tools/dp_regime.sh $TAG > $OUT/${TAG}_dp_regime.log 2>&1
bash $ROOT/tools/train_trace.sh $TAG > $OUT/${TAG}_train_trace.log 2>&1
bash $ROOT/tools/wgrad_pmc.sh $TAG > $OUT/${TAG}_wgrad_pmc.log 2>&1
cd /tmp && export TMPDIR=/tmp
unset XV_WGRAD_STREAM XV_ROUTED_POOL
TRAIN="python3 $ROOT/bench.py --mode train --steps 3 --warmup 1 --min-seconds 0 --no-cpu-baseline --no-accuracy --no-extra"
for v in train train_bn; do
  extra=""; [ $v = train_bn ] && extra="--batch-norm"
  rm -rf $OUT/${TAG}_trace_$v
  rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/${TAG}_trace_$v -o bench -- $TRAIN $extra > $OUT/${TAG}_trace_$v.log 2>&1
  f=$(ls $OUT/${TAG}_trace_$v/bench_kernel_stats.csv $OUT/${TAG}_trace_$v/*/bench_kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && cp $f $OUT/${TAG}_${v}_kernel_stats.csv
  rm -f $OUT/${TAG}_trace_$v/bench_kernel_trace.csv $OUT/${TAG}_trace_$v/*/bench_kernel_trace.csv
done
BENCH="python3 $ROOT/bench.py --steps 3 --warmup 1 --min-seconds 0 --serial-experts --no-graph --no-cpu-baseline --no-accuracy --no-extra"
rm -rf $OUT/${TAG}_trace
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/${TAG}_trace -o bench -- $BENCH > $OUT/${TAG}_trace.log 2>&1
f=$(ls $OUT/${TAG}_trace/bench_kernel_stats.csv $OUT/${TAG}_trace/*/bench_kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp $f $OUT/${TAG}_bench_serial_kernel_stats.csv
rm -f $OUT/${TAG}_trace/bench_kernel_trace.csv $OUT/${TAG}_trace/*/bench_kernel_trace.csv
AD="python3 $ROOT/bench.py --expert adapnet --steps 3 --warmup 1 --min-seconds 0 --serial-experts --no-graph --no-cpu-baseline --no-accuracy --no-extra"
rm -rf $OUT/${TAG}_trace_adapnet
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/${TAG}_trace_adapnet -o bench -- $AD > $OUT/${TAG}_trace_adapnet.log 2>&1
f=$(ls $OUT/${TAG}_trace_adapnet/bench_kernel_stats.csv $OUT/${TAG}_trace_adapnet/*/bench_kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp $f $OUT/${TAG}_adapnet_kernel_stats.csv
rm -f $OUT/${TAG}_trace_adapnet/bench_kernel_trace.csv $OUT/${TAG}_trace_adapnet/*/bench_kernel_trace.csv
tail -12 $OUT/${TAG}_records.log | cut -c1-230
for f in dirichlet_fit_2048 dp_regime adapnet adapnet_train fusion_fcn; do echo "== $f"; cut -c1-300 $OUT/${TAG}_bench_$f.json; done
cat $OUT/${TAG}_train_b4_*_gpu_busy.json | cut -c1-300
