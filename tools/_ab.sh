#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_backward_gpu.py -m gpu -x -q -k "flat_gemm" 2>&1 | tail -4
timeout 600 python -m pytest tests/test_adapnet_gpu.py -m gpu -x -q 2>&1 | tail -3
for rep in 1 2; do
  timeout 300 python bench.py --mode train --expert adapnet --batch 8 --steps 5 --warmup 2 --no-cpu-baseline --no-accuracy --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"
done
