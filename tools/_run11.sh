cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "first_pair" 2>&1 | tail -5
timeout 300 python tools/first_pair_bench.py 2>&1 | grep -v amdgpu
timeout 900 python -m pytest tests/test_models_gpu.py tests/test_fullsize_gpu.py -q -x 2>&1 | tail -4
B="python bench.py --no-cpu-baseline --no-accuracy --no-extra --steps 20 --warmup 5"
for v in 0 1 0 1; do echo "bench XV_FUSE_FIRST=$v"; XV_FUSE_FIRST=$v $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"; done
