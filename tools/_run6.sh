cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_backward_gpu.py -x -q -m gpu -k "filter_and_bias" 2>&1 | tail -5 > gpurun_out/r6f_tests.txt
for v in 2 3 2 3; do echo "== variant $v"; python3 tools/wgrad_bench.py --variant $v 2>&1 | grep -v amdgpu; done > gpurun_out/r6f_wgrad.txt
python3 -m pytest tests/test_backward_gpu.py tests/test_dp_gpu.py tests/test_fp8_gpu.py -x -q -m gpu 2>&1 | grep -v "Gloo\|amdgpu.ids\|socket" | tail -30 >> gpurun_out/r6f_tests.txt
cat gpurun_out/r6f_tests.txt | cut -c1-300; cat gpurun_out/r6f_wgrad.txt
