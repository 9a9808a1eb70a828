#!/usr/bin/env python3
"""Merge the two passes of tools/conv_power.sh into profiles/<tag>_conv_power.json: per layer the in-kernel clock and HIP-event
rate of the plain pass, the same two under the profiler, the matrix pipe's busy fraction (SQ_VALU_MFMA_BUSY_CYCLES over the
1024 SIMDs' share of GRBM_GUI_ACTIVE) and the effective clock GRBM_GUI_ACTIVE / 8 / duration of the LAST 200 launches of each
shape (the timed block), and the product busy x clock / 2.4 GHz x 2.5 PFLOP/s against the measured rate."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else 'r5'
out = os.path.join(ROOT, 'gpurun_out')
plain = json.load(open(os.path.join(out, '%s_power_clock.json' % tag)))
prof = json.load(open(os.path.join(out, '%s_power_clock_pmc.json' % tag)))
cc = glob.glob(os.path.join(out, '%s_power_pmc' % tag, '**', '*counter_collection.csv'), recursive=True)[0]
kt = glob.glob(os.path.join(out, '%s_power_pmc' % tag, '**', '*kernel_trace.csv'), recursive=True)[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r['Dispatch_Id']] = (float(r['End_Timestamp']) - float(r['Start_Timestamp'])) * 1e-9
per = {}
for r in csv.DictReader(open(cc)):
    if 'conv_dma4_kernel' not in r['Kernel_Name'] and 'conv_dma5_kernel' not in r['Kernel_Name']:
        continue
    d = per.setdefault(r['Dispatch_Id'], {'grid': r.get('Grid_Size'), 'name': r['Kernel_Name']})
    d[r['Counter_Name']] = float(r['Counter_Value'])
ids = sorted(per, key=lambda k: int(k))
rows = []
# the launches come in the order of the cases; every case ends with its 200 timed launches
n_cases = len(plain['rows'])
chunks = [ids[i * len(ids) // n_cases:(i + 1) * len(ids) // n_cases] for i in range(n_cases)]
for row, prow, chunk in zip(plain['rows'], prof['rows'], chunks):
    last = chunk[-200:]
    busy = sum(per[i]['SQ_VALU_MFMA_BUSY_CYCLES'] for i in last)
    gui = sum(per[i]['GRBM_GUI_ACTIVE'] for i in last)
    secs = sum(dur[i] for i in last if i in dur)
    util = busy / (1024.0 * gui / 8.0)
    eff_clock = gui / 8.0 / secs / 1e9 if secs > 0 else None
    flops = 2.0 * row['shape'][0] * row['shape'][1] * row['shape'][2] * row['shape'][3] * row['shape'][4] * 9
    tf_kernel = flops * len(last) / secs / 1e12 if secs > 0 else None
    rec = {'layer': row['layer'], 'cfg': row['cfg'], 'shape_n_h_w_cin_cout': row['shape'],
           'plain_pass': {'inkernel_clock_ghz': round(row['clock_ghz_median'], 3), 'loop_cycles': round(row['loop_cycles_median']),
                          'hip_event_tflops': round(row['tflops'], 1), 'us_per_launch': round(row['us_per_launch'], 2)},
           'profiled_pass': {'inkernel_clock_ghz': round(prow['clock_ghz_median'], 3), 'hip_event_tflops': round(prow['tflops'], 1),
                             'mfma_busy_fraction': round(util, 4), 'effective_clock_ghz_gui_active': round(eff_clock, 3),
                             'kernel_trace_tflops': round(tf_kernel, 1), 'launches': len(last)}}
    # busy cycles x the clock they ran at = matrix-pipe seconds; a 16x16x32 bf16 MFMA is 16 busy cycles for 16384 FLOP per SIMD
    pred = util * prow['clock_ghz_median'] / 2.4 * 2500.0
    rec['closure'] = {'busy_x_clock_over_2p4GHz_x_2500': round(pred, 1), 'kernel_trace_tflops_same_pass': round(tf_kernel, 1),
                      'ratio': round(pred / tf_kernel, 4), 'hip_event_tflops_plain_pass': round(row['tflops'], 1),
                      'note': 'MFMA-busy fraction x in-kernel clock / 2.4 GHz x 2.5 PFLOP/s against the rate of the same 200 launches '
                              'from their kernel-trace durations (the HIP-event rate of the profiled pass includes the '
                              "profiler's per-launch serialisation and is not a kernel rate); one box, one run shape (2 s of "
                              'back-to-back launches, then 200 timed ones); the plain pass on the same box gives the HIP-event rate'}
    rows.append(rec)
res = {'source': 'tools/conv_power.sh on one MI355X box in one gpurun call: tools/conv_clock.py (stamp build) plain, then under '
                 'rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES',
       'device': plain.get('device'), 'rows': rows}
os.makedirs(os.path.join(ROOT, 'profiles'), exist_ok=True)
with open(os.path.join(ROOT, 'profiles', '%s_conv_power.json' % tag), 'w') as f:
    json.dump(res, f, indent=1)
print(json.dumps(res, indent=1))
