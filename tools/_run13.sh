cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "softmax" 2>&1 | tail -2
python tools/elementwise_bench.py 2>&1 | grep -v amdgpu | python -c "
import sys,json
for r in json.loads(sys.stdin.read()):
    if 'softmax' in r['kernel']: print('%-70s %8.1f us'%(r['kernel'][:70], r['us_per_launch_events']))
"
python bench.py --mode train --batch-norm --no-cpu-baseline --no-accuracy --no-extra --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BN train', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['by_pass_tflops'])"
python bench.py --mode train --no-cpu-baseline --no-accuracy --no-extra --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['by_pass_tflops'])"
python bench.py --fusion dirichlet --no-cpu-baseline --no-accuracy --no-extra --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dirichlet', d['value'], d['ms_per_step'])"
python bench.py --no-cpu-baseline --no-accuracy --no-extra --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bayes', d['value'], d['ms_per_step'], d['roofline']['frac'])"
python bench.py --dtype fp8 --height 1024 --width 2048 --batch 4 --steps 10 --no-cpu-baseline --no-accuracy --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp8 2048', d['value'], d['ms_per_step'], d['roofline']['frac'])"
python bench.py --batch 1 --steps 50 --no-cpu-baseline --no-accuracy --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch1', d['value'], d['ms_per_step'], d['roofline']['frac'])"
