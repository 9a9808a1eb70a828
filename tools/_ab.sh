#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_models_gpu.py tests/test_backward_gpu.py -m gpu -x -q -k "fusion_fcn or joint or Fusion or fusion" 2>&1 | tail -4
for rep in 1 2; do
for f in 0 1; do
  echo "XV_ROUTED_POOL=$f"
  XV_ROUTED_POOL=$f timeout 300 python bench.py --fusion joint --mode train --steps 5 --warmup 2 --no-cpu-baseline --no-accuracy --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"
done
done
for f in 0 1; do
  echo "XV_FUSE_FIRST=$f"
  XV_FUSE_FIRST=$f timeout 300 python bench.py --fusion joint --steps 20 --warmup 5 --no-roofline-pass --no-cpu-baseline --no-accuracy --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"
done
