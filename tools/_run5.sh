cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_dp_gpu.py -x -q -m gpu -k "bucket" 2>&1 | grep -v "Gloo\|amdgpu.ids\|socket" | tail -40 > gpurun_out/r6e_tests.txt
bash tools/wgrad_exp.sh run r6 > /dev/null 2>&1
cat gpurun_out/r6e_tests.txt | cut -c1-400; cat gpurun_out/r6_wgrad_exp.txt
