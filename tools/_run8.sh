cd /tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
T="python3 $R/bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline --no-accuracy --no-extra --no-roofline-pass"
for rep in 1 2; do for v in 2 3; do
  XV_WGRAD_VARIANT=$v $T --batch 16 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('variant $v b16 plain', r['ms_per_step'], r['value'])"
  XV_WGRAD_VARIANT=$v $T --batch 16 --batch-norm 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('variant $v b16 bn   ', r['ms_per_step'], r['value'])"
done; done > $O/r6h_train_ab.txt
python3 $R/bench.py --expert adapnet --steps 10 --warmup 3 --no-cpu-baseline --no-accuracy --no-extra > $O/r6h_adapnet.json 2>/dev/null
rm -rf $O/r6h_adap_trace
rocprofv3 --output-format csv --kernel-trace --stats -d $O/r6h_adap_trace -o bench -- python3 $R/bench.py --expert adapnet --steps 3 --warmup 1 --min-seconds 0 --serial-experts --no-graph --no-cpu-baseline --no-accuracy --no-extra > $O/r6h_adap_trace.log 2>&1
cp $O/r6h_adap_trace/bench_kernel_stats.csv $O/r6h_adapnet_kernel_stats.csv 2>/dev/null || cp $O/r6h_adap_trace/*/bench_kernel_stats.csv $O/r6h_adapnet_kernel_stats.csv
rm -f $O/r6h_adap_trace/bench_kernel_trace.csv $O/r6h_adap_trace/*/bench_kernel_trace.csv
cat $O/r6h_train_ab.txt; cut -c1-1200 $O/r6h_adapnet.json; head -25 $O/r6h_adapnet_kernel_stats.csv | cut -c1-200
