#!/bin/bash
# Everything the round's committed records come from, in one GPU-box call:
#   gpurun --timeout 3000 -- 'bash tools/round_records.sh'
# then, here: copy gpurun_out/r4_* into profiles/ and run `python3 tools/pmc_summary.py r4 16`.
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -q -x -m gpu 2>&1 | tail -5
bash tools/profile_round.sh r4 > gpurun_out/r4_profile_round.log 2>&1; tail -2 gpurun_out/r4_profile_round.log
bash tools/profile_elementwise.sh r4 > gpurun_out/r4_profile_elementwise.log 2>&1; tail -3 gpurun_out/r4_profile_elementwise.log
bash tools/bench_records.sh r4 > gpurun_out/r4_bench_records.log 2>&1; tail -5 gpurun_out/r4_bench_records.log | cut -c1-160
python3 bench.py --mode train --batch-norm --no-accuracy --no-extra > gpurun_out/r4_bench_train_bn.json 2>/dev/null; cut -c1-200 gpurun_out/r4_bench_train_bn.json
