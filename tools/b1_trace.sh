cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/b1_trace
rocprofv3 --output-format csv --kernel-trace --stats -d $R/gpurun_out/b1_trace -o b1 -- python3 $R/bench.py --batch 1 --steps 50 --warmup 5 --no-cpu-baseline --no-accuracy --no-extra > $R/gpurun_out/b1_trace.log 2>&1
cd $R
python3 - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/b1_trace/b1_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:22]:
    print("%-70s %6s %9.1f us avg %5.1f%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, 100*float(r["TotalDurationNs"])/tot))
PY
