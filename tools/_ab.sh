#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_adapnet_gpu.py -m gpu -x -q 2>&1 | tail -6
for rep in 1 2; do
for f in 0 1; do
  echo "XV_IMPLICIT_PAIRS=$f"
  XV_IMPLICIT_PAIRS=$f timeout 300 python bench.py --mode train --expert adapnet --batch 8 --steps 5 --warmup 2 --no-cpu-baseline --no-accuracy --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"
done
done
