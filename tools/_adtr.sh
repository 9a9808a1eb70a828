#!/bin/bash
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out; cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/adtr
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/adtr -o t -- python3 $ROOT/bench.py --mode train --expert adapnet --batch 8 --steps 5 --warmup 2 --min-seconds 0 --no-cpu-baseline --no-accuracy --no-extra > $OUT/adtr.log 2>&1
f=$(find $OUT/adtr -name 't_kernel_stats.csv' | head -1); cp $f $OUT/r6_adapnet_train_kernel_stats.csv
find $OUT/adtr -name 't_kernel_trace.csv' -delete
head -25 $OUT/r6_adapnet_train_kernel_stats.csv | cut -c1-150
