# batch-1 step time for several chunk-group sizes of the split form (A/B on one box)
cd $GRAFT_REPO_ROOT
for g in 2 4 8 0; do
  echo "XV_COL_GROUPS=$g: $(XV_COL_GROUPS=$g python3 bench.py --batch 1 --steps 50 --warmup 5 --no-cpu-baseline --no-accuracy --no-extra 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["ms_per_step_min"])')"
done
