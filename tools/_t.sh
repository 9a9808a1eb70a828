cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_models_gpu.py -q -x -k "paired or fused_head" 2>&1 | tail -3
python bench.py --no-cpu-baseline --no-accuracy --steps 20 --warmup 5 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])
for e in d['extra'][:4]: print(e.get('workload','')[:80], e.get('value'), e.get('ms_per_step'))"
