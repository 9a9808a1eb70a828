#!/bin/bash
# Run on the GPU box (gpurun -- 'bash tools/profile_round.sh r2'): the rocprofv3 passes whose summaries are committed
# under profiles/ -- kernel trace + stats of the headline command with the experts serialised (per-kernel times), the
# FETCH_SIZE / WRITE_SIZE passes (HBM traffic of the conv kernel; separate --pmc passes, counters only), the MFMA /
# wave-cycle passes, and the same for the fp8 configuration at 2048x1024.  Raw outputs go to gpurun_out/ (scratch);
# tools/pmc_summary.py condenses them.
TAG=${1:-r2}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 3 --warmup 1 --min-seconds 0 --serial-experts --no-graph --no-cpu-baseline --no-accuracy --no-extra"
FP8="$BENCH --dtype fp8 --height 1024 --width 2048 --batch 4"
run() { d=$1; shift; rm -rf $OUT/$d; rocprofv3 --output-format csv "$@" > $OUT/$d.log 2>&1; }
run ${TAG}_trace   --kernel-trace --stats -d $OUT/${TAG}_trace -o bench -- $BENCH
run ${TAG}_trace8  --kernel-trace --stats -d $OUT/${TAG}_trace8 -o bench -- $FP8
TRAIN="python3 $ROOT/bench.py --mode train --steps 3 --warmup 1 --min-seconds 0 --no-cpu-baseline --no-accuracy --no-extra"
run ${TAG}_trace_train    --kernel-trace --stats -d $OUT/${TAG}_trace_train -o bench -- $TRAIN
run ${TAG}_trace_train_bn --kernel-trace --stats -d $OUT/${TAG}_trace_train_bn -o bench -- $TRAIN --batch-norm
run pmc_fetch      --pmc FETCH_SIZE -d $OUT/pmc_fetch -o c -- $BENCH
run pmc_write      --pmc WRITE_SIZE -d $OUT/pmc_write -o c -- $BENCH
run pmc_mfma       --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $OUT/pmc_mfma -o c -- $BENCH
run pmc_wave       --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $OUT/pmc_wave -o c -- $BENCH
run pmc_mfma8      --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $OUT/pmc_mfma8 -o c -- $FP8
run pmc_wave8      --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $OUT/pmc_wave8 -o c -- $FP8
# round 5: the label-exact mode (fp32 matrix instruction) -- kernel stats and the matrix pipe's busy fraction
EXACT="python3 $ROOT/tools/exact_bench.py 16 3"
run ${TAG}_trace_exact --kernel-trace --stats -d $OUT/${TAG}_trace_exact -o bench -- $EXACT
run pmc_mfma_exact --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $OUT/pmc_mfma_exact -o c -- $EXACT
run pmc_fetch_exact --pmc FETCH_SIZE -d $OUT/pmc_fetch_exact -o c -- $EXACT
run pmc_write_exact --pmc WRITE_SIZE -d $OUT/pmc_write_exact -o c -- $EXACT
# the elementwise / reduction kernels: event timing, then one launch each under the counters
EW="python3 $ROOT/tools/elementwise_bench.py"
$EW > $OUT/${TAG}_ew_timing.json 2>/dev/null
run ew_fetch --pmc FETCH_SIZE -d $OUT/ew_fetch -o c -- $EW --once
run ew_write --pmc WRITE_SIZE -d $OUT/ew_write -o c -- $EW --once
run ew_trace --kernel-trace --stats -d $OUT/ew_trace -o c -- $EW --once
cd $ROOT
python3 tools/elementwise_bench.py --tag $TAG --merge $OUT/ew_fetch $OUT/ew_write $OUT/ew_trace < $OUT/${TAG}_ew_timing.json
python3 tools/pmc_summary.py $TAG 16
ls $OUT/${TAG}_trace/*/ 2>/dev/null | head; tail -3 $OUT/${TAG}_trace.log
