cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_backward_gpu.py -x -q -m gpu -k "filter_and_bias" 2>&1 | tail -3 > gpurun_out/r6g_tests.txt
for v in 2 3 2 3; do echo "== variant $v"; python3 tools/wgrad_bench.py --variant $v 2>&1 | grep -v amdgpu; done > gpurun_out/r6g_wgrad.txt
python3 -m pytest tests/test_backward_gpu.py tests/test_dp_gpu.py tests/test_fp8_gpu.py tests/test_models_gpu.py -x -q -m gpu 2>&1 | grep -v "Gloo\|amdgpu.ids\|socket" | tail -30 >> gpurun_out/r6g_tests.txt
cd /tmp; for b in 16 4; do python3 $GRAFT_REPO_ROOT/bench.py --mode train --batch $b --steps 10 --warmup 3 --no-cpu-baseline --no-accuracy --no-extra > $GRAFT_REPO_ROOT/gpurun_out/r6g_train_b$b.json 2>/dev/null; done
cd $GRAFT_REPO_ROOT
cat gpurun_out/r6g_tests.txt | cut -c1-300; cat gpurun_out/r6g_wgrad.txt; cut -c1-900 gpurun_out/r6g_train_b16.json; cut -c1-400 gpurun_out/r6g_train_b4.json
