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
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONV_SRC = os.path.join(ROOT, 'modular_semantic_segmentation_amd', 'csrc', 'conv_mfma.hip')


def kernel_source_hash():
    """Ties a committed counter summary to the kernel source it was measured on: bench.py reports `traffic` only
    while this still matches."""
    return hashlib.sha256(open(CONV_SRC, 'rb').read()).hexdigest()[:16]


def counter_avg(dirname, counter, match):
    f = glob.glob('gpurun_out/%s/*/*counter_collection.csv' % dirname)
    if not f:
        return None
    n, tot = 0, 0.0
    for r in csv.DictReader(open(f[0])):
        if r['Counter_Name'] == counter and match(r['Kernel_Name']):
            n += 1
            tot += float(r['Counter_Value'])
    return (tot / n, n) if n else None


def main(tag, batch=16):
    is_conv3 = lambda k: 'conv_dma_kernel' in k or ('conv_mfma_kernel' in k and ', 3, ' in k)      # noqa: E731
    fetch = counter_avg('pmc_fetch', 'FETCH_SIZE', is_conv3)
    write = counter_avg('pmc_write', 'WRITE_SIZE', is_conv3)
    out = {'kernel': 'conv_dma_kernel + conv_mfma_kernel (all 3x3 launches)', 'source': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, '
           'python3 bench.py --steps 3 --warmup 1 --serial-experts --no-graph (batch %d, 768x384)' % batch,
           'batch': batch, 'kernel_source': 'csrc/conv_mfma.hip', 'kernel_source_sha256_16': kernel_source_hash()}
    if fetch and write:
        out.update(fetch_size_kb_per_launch=round(fetch[0], 1), write_size_kb_per_launch=round(write[0], 1),
                   launches=fetch[1],
                   hbm_bytes_per_launch=int((2 * fetch[0] + write[0]) * 1024),
                   correction='read bytes = 2 x FETCH_SIZE (gfx950 counts 128-B requests as 64 B); WRITE_SIZE exact')
    json.dump(out, open('profiles/%s_conv_traffic.json' % tag, 'w'), indent=1)
    print(out)


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else 'r1', int(sys.argv[2]) if len(sys.argv) > 2 else 16)
