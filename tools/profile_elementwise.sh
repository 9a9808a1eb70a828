#!/bin/bash
# GPU box: gpurun -- 'bash tools/profile_elementwise.sh r3'.  Event timing + rocprofv3 kernel stats + separate FETCH_SIZE /
# WRITE_SIZE passes (counters only) of every elementwise kernel; summary -> profiles/<tag>_elementwise.json.
TAG=${1:-r3}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/tools/elementwise_bench.py > $OUT/ew_timing.json 2> $OUT/ew_timing.err || { tail -5 $OUT/ew_timing.err; exit 1; }
run() { d=$1; shift; rm -rf $OUT/$d; rocprofv3 --output-format csv "$@" > $OUT/$d.log 2>&1; }
run ew_trace --kernel-trace --stats -d $OUT/ew_trace -o ew -- python3 $ROOT/tools/elementwise_bench.py --once
run ew_fetch --pmc FETCH_SIZE -d $OUT/ew_fetch -o ew -- python3 $ROOT/tools/elementwise_bench.py --once
run ew_write --pmc WRITE_SIZE -d $OUT/ew_write -o ew -- python3 $ROOT/tools/elementwise_bench.py --once
cd $ROOT
python3 tools/elementwise_bench.py --tag $TAG --merge $OUT/ew_fetch $OUT/ew_write $OUT/ew_trace < $OUT/ew_timing.json
cp profiles/${TAG}_elementwise.json $OUT/
python3 -c "
import json
for r in json.load(open('profiles/${TAG}_elementwise.json'))['kernels']:
    print('%-62s %8.1f us %7.0f GB/s %.3f  traffic x%s' % (r['kernel'][:62], r.get('us_per_launch_events', 0), r.get('achieved_GBps', 0), r.get('frac_of_8TBps', 0), r.get('traffic_over_algorithmic')))
"
