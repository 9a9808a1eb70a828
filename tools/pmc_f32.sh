cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_f32
rocprofv3 --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --stats -d $R/gpurun_out/pmc_f32 -o f32 -- python3 $R/tools/exact_bench.py 16 2 > $R/gpurun_out/pmc_f32.log 2>&1
rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d $R/gpurun_out/pmc_f32w -o f32 -- python3 $R/tools/exact_bench.py 16 2 > $R/gpurun_out/pmc_f32w.log 2>&1
cd $R
python3 - <<PY
import csv,glob
for d in ("pmc_f32","pmc_f32w"):
    f=glob.glob("gpurun_out/%s/**/*counter_collection.csv"%d,recursive=True)
    acc={}
    for r in csv.DictReader(open(f[0])):
        k=r["Kernel_Name"]
        if "conv_f32_mfma" in k and ("3, 16" in k or "Li3ELi16" in k):
            d2=acc.setdefault(r["Counter_Name"],[0,0]); d2[0]+=float(r["Counter_Value"]); d2[1]+=1
    print(d, {k:(v[0]/v[1], v[1]) for k,v in acc.items()})
g=glob.glob("gpurun_out/pmc_f32/**/*kernel_stats.csv",recursive=True)
print(open(g[0]).read()[:1200] if g else "nostats")
PY
