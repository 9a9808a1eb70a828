cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_dp_gpu.py -x -q -m gpu -k "bucket" 2>&1 | tail -15 > gpurun_out/r6d_tests.txt
bash tools/wgrad_pmc.sh r6 > gpurun_out/r6d_pmc.txt 2>&1
cat gpurun_out/r6d_tests.txt; tail -130 gpurun_out/r6d_pmc.txt
