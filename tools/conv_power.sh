#!/bin/bash
# MFMA-busy fraction, in-kernel clock and HIP-event rate of the generation-4 conv FROM ONE BOX AND ONE RUN
# SHAPE.  Run on the GPU box (gpurun -- 'bash tools/conv_power.sh r5'): the same script (tools/conv_clock.py on the diagnostic
# stamp build: 2 s of back-to-back launches of ONE layer, then 200 timed launches whose last one leaves the clock stamps) runs
# twice in this call -- plain (in-kernel clock, loop cycles, HIP-event TFLOP/s) and under rocprofv3 --pmc (MFMA busy cycles,
# GRBM_GUI_ACTIVE, with the kernel trace for durations) -- and tools/conv_power_merge.py writes profiles/<tag>_conv_power.json.
TAG=${1:-r5}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
CASES=${2:-conv3_2:26,conv4_2:26}
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/tools/conv_clock.py --cases $CASES --data normal --seconds 2 --out $OUT/${TAG}_power_clock.json > $OUT/${TAG}_power_clock.txt 2>&1
rm -rf $OUT/${TAG}_power_pmc
rocprofv3 --output-format csv --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $OUT/${TAG}_power_pmc -o p -- \
  python3 $ROOT/tools/conv_clock.py --cases $CASES --data normal --seconds 2 --out $OUT/${TAG}_power_clock_pmc.json > $OUT/${TAG}_power_pmc.txt 2>&1
cd $ROOT
python3 tools/conv_power_merge.py $TAG
