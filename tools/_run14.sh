cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -q -x -m gpu 2>&1 | tail -6
timeout 600 python tools/conv_fuzz.py --cases 90 2>&1 | tail -2
timeout 600 python tools/stress_cold.py --iters 4 2>&1 | tail -2
