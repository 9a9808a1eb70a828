cd $GRAFT_REPO_ROOT
python bench.py --batch 1 --steps 50 --no-cpu-baseline --no-accuracy --no-extra --layer-profile --serial-experts --no-graph 2>&1 | grep -E "conv launch" | head -14
python bench.py --batch 1 --steps 50 --no-cpu-baseline --no-accuracy --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['measured'])"
