#!/usr/bin/env python3
"""Summarise rocprofv3 outputs (run on the GPU box into gpurun_out/) into small committed files
under profiles/: per-kernel time stats and per-launch HBM traffic of the conv kernel from the
FETCH_SIZE / WRITE_SIZE passes (separate --pmc passes; gfx950 correction: FETCH_SIZE reports half
the bytes of wide coalesced reads -> doubled, WRITE_SIZE exact; MI355X_MICROARCH.md 'HBM')."""
import csv
import glob
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONV_SRCS = [os.path.join(ROOT, 'modular_semantic_segmentation_amd', 'csrc', f) for f in ('conv_mfma.hip', 'conv_f8_dma.hip', 'conv_col_dma.hip', 'conv_first_fused.hip')]


def kernel_source_hash():
    """Ties a committed counter summary to the kernel source it was measured on: bench.py reports `traffic` only
    while this still matches."""
    return hashlib.sha256(b''.join(open(f, 'rb').read() for f in CONV_SRCS)).hexdigest()[:16]


def counter_avg(dirname, counter, match):
    f = glob.glob('gpurun_out/%s/**/*counter_collection.csv' % dirname, recursive=True)
    if not f:
        return None
    n, tot = 0, 0.0
    for r in csv.DictReader(open(f[0])):
        if r['Counter_Name'] == counter and match(r['Kernel_Name']):
            n += 1
            tot += float(r['Counter_Value'])
    return (tot / n, n) if n else None


def counters_by_kernel(dirname, groups):
    """{group: {counter: (mean per launch, launches)}} from one --pmc pass; groups = {name: predicate on kernel name}."""
    f = glob.glob('gpurun_out/%s/**/*counter_collection.csv' % dirname, recursive=True)
    out = {}
    if not f:
        return out
    acc = {}
    for r in csv.DictReader(open(f[0])):
        for g, match in groups.items():
            if match(r['Kernel_Name']):
                d = acc.setdefault((g, r['Counter_Name']), [0.0, 0])
                d[0] += float(r['Counter_Value'])
                d[1] += 1
    for (g, c), (tot, n) in acc.items():
        out.setdefault(g, {})[c] = (tot / n, n)
    return out


def mfma_summary(tag, suffix, groups, command):
    """MFMA utilisation and the wave-cycle split per kernel group (gfx94x MfmaUtil formula: matrix-pipe busy cycles
    summed over the 1024 SIMDs over GRBM_GUI_ACTIVE summed over the 8 XCDs)."""
    mf = counters_by_kernel('pmc_mfma' + suffix, groups)
    wv = counters_by_kernel('pmc_wave' + suffix, groups)
    out = {'source': 'rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES -- ' + command +
           ' (counters only, no traces); second pass: SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY '
           'SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS',
           'note': 'mfma_util = MFMA_BUSY / (1024 * GUI_ACTIVE / 8); wave-cycle fractions are of SQ_WAVE_CYCLES (WAIT_ANY = parked '
                   'at s_waitcnt / s_barrier, WAIT_INST_ANY = issue stall, mostly the matrix pipe busy with the SIMD\'s other wave)',
           'kernel_source_sha256_16': kernel_source_hash(), 'kernels': {}}
    for g in groups:
        if g not in mf or 'SQ_VALU_MFMA_BUSY_CYCLES' not in mf[g]:
            continue
        busy, n = mf[g]['SQ_VALU_MFMA_BUSY_CYCLES']
        gui = mf[g]['GRBM_GUI_ACTIVE'][0]
        rec = {'launches': n, 'mfma_busy_cycles_per_launch': int(busy), 'gui_active_cycles_per_launch': int(gui),
               'mfma_util': round(busy / (1024 * gui / 8), 4)}
        if g in wv and 'SQ_WAVE_CYCLES' in wv[g]:
            tot = wv[g]['SQ_WAVE_CYCLES'][0]
            rec['wave_cycle_fractions'] = {c: round(v[0] / tot, 3) for c, v in sorted(wv[g].items()) if c != 'SQ_WAVE_CYCLES'}
        out['kernels'][g] = rec
    json.dump(out, open('profiles/%s_conv_mfma_util%s.json' % (tag, '_fp8' if suffix else ''), 'w'), indent=1)
    print(out)


def exact_summary(tag):
    """conv_dtype='fp32' (tools/exact_bench.py 16 3): the fp32-MFMA conv kernel's matrix-pipe busy fraction and HBM traffic."""
    is_k = lambda k: 'conv_f32_mfma_kernel<3, 16' in k          # noqa: E731
    mf = counters_by_kernel('pmc_mfma_exact', {'k': is_k}).get('k', {})
    f, w = counter_avg('pmc_fetch_exact', 'FETCH_SIZE', is_k), counter_avg('pmc_write_exact', 'WRITE_SIZE', is_k)
    if not mf:
        return
    busy, gui = mf['SQ_VALU_MFMA_BUSY_CYCLES'][0], mf['GRBM_GUI_ACTIVE'][0]
    out = {'kernel': 'conv_f32_mfma_kernel<3, 16, true> (v_mfma_f32_32x32x2_f32: conv1_2 .. conv5_3 of both experts, 16 images)',
           'source': 'rocprofv3 --pmc ... -- python3 tools/exact_bench.py 16 3 (separate passes: MFMA / FETCH_SIZE / WRITE_SIZE)',
           'launches': mf['SQ_VALU_MFMA_BUSY_CYCLES'][1], 'mfma_busy_cycles_per_launch': int(busy),
           'gui_active_cycles_per_launch': int(gui), 'mfma_util': round(busy / (1024 * gui / 8), 4),
           'note': 'a 32x32x2 f32 MFMA holds its SIMD 64 cycles for 4096 FLOP: 157.3 TFLOP/s at 2.4 GHz'}
    if f and w:
        out.update(fetch_size_kb_per_launch=round(f[0], 1), write_size_kb_per_launch=round(w[0], 1),
                   hbm_bytes_per_launch=int((2 * f[0] + w[0]) * 1024),
                   correction='read bytes = 2 x FETCH_SIZE (gfx950 counts 128-B requests as 64 B); WRITE_SIZE exact')
    json.dump(out, open('profiles/%s_exact_conv_counters.json' % tag, 'w'), indent=1)
    print(out)


def copy_stats(tag, sub, dest):
    """The rocprofv3 --kernel-trace --stats summary (per-kernel calls / total / average ns) -> profiles/."""
    f = glob.glob('gpurun_out/%s/**/*kernel_stats.csv' % sub, recursive=True)
    if f:
        open('profiles/%s' % dest, 'w').write(open(f[0]).read())
        print('copied', f[0], '->', dest)


def main(tag, batch=16):
    # the plain 3x3 conv kernels = what bench.py's `roofline` record is about (the fused first pair is reported beside it)
    is_conv3 = lambda k: ('conv_dma_kernel' in k or 'conv_dma4_kernel' in k or 'conv_dma5_kernel' in k  # noqa: E731
                          or ('conv_mfma_kernel' in k and ', 3, ' in k))
    fetch = counter_avg('pmc_fetch', 'FETCH_SIZE', is_conv3)
    write = counter_avg('pmc_write', 'WRITE_SIZE', is_conv3)
    out = {'kernel': 'conv_dma4_kernel + conv_dma5_kernel (+ conv_dma_kernel / conv_mfma_kernel): the plain 3x3 launches', 'source': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, '
           'python3 bench.py --steps 3 --warmup 1 --serial-experts --no-graph (batch %d, 768x384)' % batch,
           'batch': batch, 'kernel_source': 'csrc/conv_mfma.hip + conv_f8_dma.hip + conv_col_dma.hip + conv_first_fused.hip', 'kernel_source_sha256_16': kernel_source_hash()}
    if fetch and write:
        out.update(fetch_size_kb_per_launch=round(fetch[0], 1), write_size_kb_per_launch=round(write[0], 1),
                   launches=fetch[1],
                   hbm_bytes_per_launch=int((2 * fetch[0] + write[0]) * 1024),
                   correction='read bytes = 2 x FETCH_SIZE (gfx950 counts 128-B requests as 64 B); WRITE_SIZE exact')
    pf = counter_avg('pmc_fetch', 'FETCH_SIZE', lambda k: 'conv_first_pair_kernel' in k)
    pw = counter_avg('pmc_write', 'WRITE_SIZE', lambda k: 'conv_first_pair_kernel' in k)
    if pf and pw:
        out['conv_first_pair_kernel'] = {'fetch_size_kb_per_launch': round(pf[0], 1), 'write_size_kb_per_launch': round(pw[0], 1),
                                         'launches': pf[1], 'hbm_bytes_per_launch': int((2 * pf[0] + pw[0]) * 1024)}
    json.dump(out, open('profiles/%s_conv_traffic.json' % tag, 'w'), indent=1)
    print(out)
    def is_f8(k):
        """conv_mfma_kernel<MT, WR, WC, NW, KS, OCC, TPS, DMAB, F8>: the last template argument"""
        m = re.search(r'conv_mfma_kernel<([^>]*)>', k)
        return bool(m) and m.group(1).split(',')[-1].strip() in ('1', '2')

    cmd = 'python3 bench.py --steps 3 --warmup 1 --serial-experts --no-graph --no-cpu-baseline --no-accuracy --no-extra'
    mfma_summary(tag, '', {'conv_dma_kernel': lambda k: 'conv_dma_kernel' in k,
                           'conv_dma4_kernel<false, false, ..., M16> (generation 4 on 16x16x32, bf16)': lambda k: 'conv_dma4_kernel<false, false' in k,
                           'conv_dma5_kernel<3> (generation 5, 24x16 tiles: conv5_x)': lambda k: 'conv_dma5_kernel<3' in k,
                           'conv_first_pair_kernel (conv1_1 + conv1_2 + pool1 fused)': lambda k: 'conv_first_pair_kernel' in k,
                           'conv_mfma_kernel (bf16, first generation)': lambda k: 'conv_mfma_kernel' in k and not is_f8(k)},
                 cmd + ' (batch 16, 768x384)')
    mfma_summary(tag, '8', {'conv_dma4_kernel<true, true> (generation 4, e4m3 operands)': lambda k: 'conv_dma4_kernel<true' in k,
                            'conv_dma4_kernel<false, true> (conv1_2: bf16 in, e4m3 out)': lambda k: 'conv_dma4_kernel<false, true' in k,
                            'conv_mfma_kernel<F8> (e4m3 operands, first generation)': is_f8,
                            'conv_dma_kernel (bf16)': lambda k: 'conv_dma_kernel' in k},
                 cmd + ' --dtype fp8 --height 1024 --width 2048 --batch 4')
    exact_summary(tag)
    copy_stats(tag, tag + '_trace_exact', '%s_exact_kernel_stats.csv' % tag)
    copy_stats(tag, tag + '_trace', '%s_bench_serial_kernel_stats.csv' % tag)
    copy_stats(tag, tag + '_trace8', '%s_bench_fp8_2048_kernel_stats.csv' % tag)
    copy_stats(tag, tag + '_trace_train', '%s_train_kernel_stats.csv' % tag)
    copy_stats(tag, tag + '_trace_train_bn', '%s_train_bn_kernel_stats.csv' % tag)


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else 'r1', int(sys.argv[2]) if len(sys.argv) > 2 else 16)
