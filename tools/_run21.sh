cd $GRAFT_REPO_ROOT
timeout 900 python tools/stress_determinism.py 2>&1 | grep -v amdgpu | tail -4
timeout 900 python tools/stress_cold.py --iters 60 2>&1 | grep -v amdgpu | tail -3
timeout 900 python tools/stress_train_step.py --iters 30 --cold 2>&1 | grep -v amdgpu | tail -4
timeout 600 python tools/conv_fuzz.py --cases 300 --seed 11 2>&1 | tail -2
timeout 600 python tools/conv_fuzz.py --fp8 --cases 150 --seed 5 2>&1 | tail -2
