cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu -k "first_pair or fullsize or full_size or 768" 2>&1 | tail -3
python3 bench.py --no-cpu-baseline --no-accuracy --no-extra 2>/dev/null | cut -c1-400
