cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/test_batchnorm_gpu.py tests/test_adapnet_gpu.py tests/test_fp8_gpu.py tests/test_backward_gpu.py tests/test_frozen_state_gradients_gpu.py tests/test_dp_gpu.py -q -x 2>&1 | tail -4
bash tools/profile_round.sh r4 > gpurun_out/r4_profile_round.log 2>&1; tail -3 gpurun_out/r4_profile_round.log
bash tools/profile_elementwise.sh r4 > gpurun_out/r4_profile_elementwise.log 2>&1; tail -20 gpurun_out/r4_profile_elementwise.log
bash tools/bench_records.sh r4 > gpurun_out/r4_bench_records.log 2>&1; tail -5 gpurun_out/r4_bench_records.log | cut -c1-200
python3 bench.py --mode train --batch-norm --no-accuracy --no-extra > gpurun_out/r4_bench_train_bn.json 2>/dev/null; cut -c1-300 gpurun_out/r4_bench_train_bn.json
