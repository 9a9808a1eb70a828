cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_models_gpu.py -q -x -k "dirichlet or fused or fusion" 2>&1 | tail -3
timeout 600 python tools/elementwise_bench.py 2>&1 | grep -v amdgpu | python -c "
import sys,json
for r in json.loads(sys.stdin.read()):
    print('%-70s %8.1f us'%(r['kernel'][:70], r['us_per_launch_events']))
"
