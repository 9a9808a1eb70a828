#!/bin/bash
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_adapnet_gpu.py -m gpu -x -q 2>&1 | tail -8
for rep in 1 2 3; do
  timeout 300 python bench.py --expert adapnet --steps 30 --warmup 5 --no-roofline-pass --no-cpu-baseline --no-accuracy --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"
done
