#!/bin/bash
# GPU box: counters of the filter-gradient kernel inside the running training step (one stream, so a launch has the chip to
# itself): matrix-pipe busy cycles, LDS array cycles and bank-conflict cycles, wave-cycle split.  Two --pmc passes, kernel
# trace for durations.   gpurun -- 'bash tools/wgrad_pmc.sh r6'
TAG=${1:-r6}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
export XV_WGRAD_STREAM=0
TRAIN="python3 $ROOT/bench.py --mode train --steps 3 --warmup 1 --min-seconds 0 --no-cpu-baseline --no-accuracy --no-extra --no-roofline-pass"
rm -rf $OUT/${TAG}_wgrad_pmc1 $OUT/${TAG}_wgrad_pmc2
rocprofv3 --output-format csv --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $OUT/${TAG}_wgrad_pmc1 -o p -- $TRAIN > $OUT/${TAG}_wgrad_pmc1.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS -d $OUT/${TAG}_wgrad_pmc2 -o p -- $TRAIN > $OUT/${TAG}_wgrad_pmc2.log 2>&1
python3 - <<PY > $OUT/${TAG}_wgrad_counters.json
import csv, glob, json
out = {}
for d in ('${TAG}_wgrad_pmc1', '${TAG}_wgrad_pmc2'):
    for f in glob.glob('$OUT/%s/**/*counter_collection.csv' % d, recursive=True):
        acc = {}
        for r in csv.DictReader(open(f)):
            name = r['Kernel_Name']
            key = None
            for k in ('conv_wgrad_lw_kernel', 'conv_wgrad_dma_kernel', 'conv_dma4_kernel', 'conv_dma5_kernel'):
                if k in name:
                    key = k + ('<DG>' if k == 'conv_dma4_kernel' and 'true, true, false>' in name.replace(', true>', ', true, X>') else '')
            if key is None:
                continue
            a = acc.setdefault((key, r['Counter_Name']), [0.0, 0])
            a[0] += float(r['Counter_Value']); a[1] += 1
        for (k, c), (tot, n) in acc.items():
            out.setdefault(k, {})[c] = {'mean_per_launch': tot / n, 'launches': n}
for k, c in out.items():
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in c and 'GRBM_GUI_ACTIVE' in c:
        # (gfx94x MfmaUtil: matrix-pipe busy cycles over the 1024 SIMDs / GRBM_GUI_ACTIVE over the 8 XCDs)
        c['mfma_busy_frac'] = c['SQ_VALU_MFMA_BUSY_CYCLES']['mean_per_launch'] / (c['GRBM_GUI_ACTIVE']['mean_per_launch'] / 8 * 1024)
    if 'SQ_LDS_BANK_CONFLICT' in c and 'SQ_LDS_IDX_ACTIVE' in c and c['SQ_LDS_IDX_ACTIVE']['mean_per_launch'] > 0:
        c['lds_conflict_frac_of_lds_cycles'] = c['SQ_LDS_BANK_CONFLICT']['mean_per_launch'] / c['SQ_LDS_IDX_ACTIVE']['mean_per_launch']
print(json.dumps(out, indent=1))
PY
cat $OUT/${TAG}_wgrad_counters.json | head -120
