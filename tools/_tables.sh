#!/bin/bash
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out; cd /tmp && export TMPDIR=/tmp
run() { tag=$1; shift; rm -rf $OUT/tb_$tag; rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/tb_$tag -o t -- python3 $ROOT/bench.py "$@" --min-seconds 0 --no-cpu-baseline --no-accuracy --no-extra > $OUT/tb_$tag.log 2>&1; f=$(find $OUT/tb_$tag -name 't_kernel_stats.csv' | head -1); cp $f $OUT/tb_${tag}_kernel_stats.csv; find $OUT/tb_$tag -name 't_kernel_trace.csv' -delete; grep '^{' $OUT/tb_$tag.log | tail -1 | cut -c1-200; }
run joint_infer --fusion joint --steps 5 --warmup 2 --no-roofline-pass
run joint_train --fusion joint --mode train --steps 5 --warmup 2
