#!/usr/bin/env python3
"""HBM-side evidence for the elementwise / reduction kernels of the hot path (north_star: "the softmax + Bayes/Dirichlet
mix + argmax as wavefront-reduced elementwise kernels, the choices evidenced by rocprof HBM GB/s"): every such kernel is
launched alone on synthetic operands of the headline size (16 images of 768x384, C = 12, U = 64) and timed with HIP
events on its launch stream; its ALGORITHMIC bytes (what the op must read and write once) over
that time is the achieved rate, against the 8 TB/s HBM3E peak (6.29 TB/s measured for a float4 copy,
MI355X_MICROARCH.md).

  python3 tools/elementwise_bench.py                  -> one JSON record per kernel on stdout (event timing)
  rocprofv3 --pmc FETCH_SIZE ... -- python3 tools/elementwise_bench.py --once     (counter passes: one launch each)
  python3 tools/elementwise_bench.py --merge gpurun_out/ew_fetch gpurun_out/ew_write gpurun_out/ew_trace < timing.json

--merge adds, per kernel, the rocprofv3 average duration and the HBM traffic of one launch from the FETCH_SIZE /
WRITE_SIZE passes (gfx950 correction of the guide: FETCH_SIZE counts 64 B per 128-B request of a wide coalesced read
-> doubled; WRITE_SIZE exact) and writes profiles/<tag>_elementwise.json.
"""
import argparse
import csv
import glob
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
N, H, W, C, U = 16, 384, 768, 12, 64
PEAK_GBS, COPY_GBS = 8000.0, 6290.0


def cases(ops, dev):
    """(name, kernel-name substring for rocprof, algorithmic bytes per launch, callable)"""
    g = torch.Generator(device='cpu').manual_seed(0)
    npix = N * H * W
    out = []
    la = torch.randint(0, C, (N, H, W), generator=g).to(dev)
    lb = torch.randint(0, C, (N, H, W), generator=g).to(dev)
    loglik = torch.randn((2, C, C), generator=g).to(dev)
    logprior = torch.randn(C, generator=g).to(dev)
    out.append(('bayes_fuse (labels -> fused label)', 'bayes_fuse_kernel', npix * (8 + 8 + 8),
                lambda: ops.bayes_fuse([la, lb], loglik, logprior)))
    lut = torch.randint(0, C, (C, C), generator=g).to(dev)
    out.append(('bayes_fuse_lut', 'bayes_lut_kernel', npix * 24, lambda: ops.bayes_fuse_lut(la, lb, lut)))
    pa = torch.softmax(torch.randn((N, H, W, C), generator=g), -1).to(dev)
    pb = torch.softmax(torch.randn((N, H, W, C), generator=g), -1).to(dev)
    am1 = torch.rand((2, C, C), generator=g).to(dev)
    lognorm = torch.randn((2, C), generator=g).to(dev)
    out.append(('dirichlet_fuse (probabilities -> fused label)', 'dirichlet_fuse_pk_kernel', npix * (2 * C * 4 + 8),
                lambda: ops.dirichlet_fuse([pa, pb], am1, lognorm, logprior)))
    out.append(('average_fuse', 'average_fuse_kernel', npix * (2 * C * 4 + 8), lambda: ops.average_fuse([pa, pb])))
    score = torch.randn((N, H, W, C), generator=g).to(dev)
    out.append(('softmax_argmax (score -> prob + label)', 'softmax_argmax_kernel', npix * (C * 4 + C * 4 + 8),
                lambda: ops.softmax_argmax(score)))
    out.append(('softmax_argmax (score -> label only)', 'softmax_argmax_kernel', npix * (C * 4 + 8),
                lambda: ops.softmax_argmax(score, want_prob=False)))
    labels = torch.randint(-1, C, (N, H, W), generator=g).int().to(dev)
    S = torch.zeros((C, C), dtype=torch.float64, device=dev)
    counts = torch.zeros(C, dtype=torch.int64, device=dev)
    out.append(('dirichlet_suffstats', 'suffstats_kernel', npix * (C * 4 + 4),
                lambda: ops.dirichlet_suffstats(pa, labels, S, counts)))
    cm = torch.zeros((C, C), dtype=torch.int64, device=dev)
    out.append(('confusion_matrix', 'confusion_kernel', npix * (4 + 8), lambda: ops.confusion_matrix(labels, la, cm)))
    # decoder side: 1/8-resolution features -> full-resolution label map
    hi, wi = H // 8, W // 8
    fused = ops.Act.from_dense(torch.rand((N, hi, wi, U), generator=g).to(dev))
    ws = torch.randn((U, C), generator=g).to(dev)
    bs = torch.randn(C, generator=g).to(dev)
    cp = (C + 3) // 4 * 4
    Sa = torch.zeros((N, hi + 2, wi + 2, cp), device=dev)
    Sb = torch.zeros((N, hi + 2, wi + 2, cp), device=dev)
    ops.score_lowres(fused, ws, C, Sa)
    ops.score_lowres(fused, ws, C, Sb)
    out.append(('score_lowres (1/8-resolution class scores)', 'score_lowres_kernel', N * hi * wi * (U * 2 + cp * 4),
                lambda: ops.score_lowres(fused, ws, C, Sa)))
    tab = torch.randn((2, C, C), generator=g).to(dev)
    lab_out = torch.empty((N, H, W), dtype=torch.int64, device=dev)
    low_bytes = 2 * N * (hi + 2) * (wi + 2) * cp * 4
    out.append(('fused_head, Bayes (both experts\' low-res scores -> fused label)', 'fused_head_kernel<12, 0,', npix * 8 + low_bytes,
                lambda: ops.fused_head(Sa, Sb, bs, bs, N, hi, wi, C, tab, logprior, out=lab_out)))
    out.append(('fused_head, Dirichlet', 'fused_dirichlet_head_pk_kernel<12', npix * 8 + low_bytes,
                lambda: ops.fused_head(Sa, Sb, bs, bs, N, hi, wi, C, am1, logprior, lognorm=lognorm, out=lab_out)))
    head_out = {}
    out.append(('decoder_head (one expert: features -> label)', 'decoder_head_kernel', npix * 8 + N * hi * wi * U * 2,
                lambda: ops.decoder_head_fwd(fused, ws, bs, C, out=head_out)))
    s5 = ops.Act.from_dense(torch.rand((N, hi // 2, wi // 2, U), generator=g).to(dev))
    res = ops.Act.from_dense(torch.rand((N, hi, wi, U), generator=g).to(dev))
    yy = ops.Act(N, hi, wi, U, dev)
    out.append(('upsample2x_relu_add', 'upsample2x_kernel', N * hi * wi * U * 2 * 2 + N * (hi // 2) * (wi // 2) * U * 2,
                lambda: ops.upsample2x_relu_add(s5, residual=res, y=yy)))
    # training side
    y1 = ops.Act.from_dense(torch.rand((N, H // 2, W // 2, 128), generator=g).to(dev))
    dq = ops.Act.from_dense(torch.rand((N, H // 4, W // 4, 128), generator=g).to(dev))
    dy = ops.Act(N, H // 2, W // 2, 128, dev)
    nb = N * (H // 2) * (W // 2) * 128 * 2
    out.append(('maxpool2x2_bwd (pool2)', 'maxpool_bwd_kernel', nb + nb + nb // 4, lambda: ops.maxpool2x2_bwd(y1, dq, dy)))
    q = ops.Act(N, H // 4, W // 4, 128, dev)
    out.append(('maxpool2x2_fwd (stand-alone form)', 'maxpool_kernel', nb + nb // 4, lambda: ops.maxpool2x2_fwd(y1, q)))
    nparam = 14781132
    p, gr, m, v = (torch.randn(nparam, generator=g).to(dev) for _ in range(4))
    v.abs_()
    out.append(('adam_step (one expert, 14.78 M parameters)', 'adam_kernel', nparam * (4 * 4 + 3 * 4),
                lambda: ops.adam_step(p, gr, m, v, 1e-4)))
    x = torch.randint(0, 256, (N, H, W, 3), generator=g).float().to(dev)
    w1 = torch.randn((3, 3, 3, 64), generator=g).to(dev) * 0.02
    b1 = torch.zeros(64, device=dev)
    y0 = ops.Act(N, H, W, 64, dev)
    out.append(('conv1_1 (fp32 RGB in, bf16 64-channel map out)', 'conv_first_mfma_kernel<3>', npix * (3 * 4 + 64 * 2),
                lambda: ops.conv2d_first_fwd(x, w1, b1, y0)))
    return out


def measure(iters):
    from modular_semantic_segmentation_amd import ops
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    recs = []
    for name, kern, nbytes, fn in cases(ops, dev):
        fn()
        torch.cuda.synchronize()
        if iters <= 1:
            recs.append({'kernel': name, 'rocprof_name': kern, 'algorithmic_bytes': int(nbytes)})
            continue
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters
        gbs = nbytes / us / 1e3
        recs.append({'kernel': name, 'rocprof_name': kern, 'algorithmic_bytes': int(nbytes), 'us_per_launch_events': round(us, 2),
                     'achieved_GBps': round(gbs, 1), 'frac_of_8TBps': round(gbs / PEAK_GBS, 4),
                     'frac_of_measured_copy_6.29TBps': round(gbs / COPY_GBS, 4)})
    return recs


def merge(tag, fetch_dir, write_dir, trace_dir):
    recs = json.load(sys.stdin)

    def counter(dirname, cname):
        out = {}
        for f in glob.glob(os.path.join(dirname, '**', '*counter_collection.csv'), recursive=True):
            for r in csv.DictReader(open(f)):
                if r['Counter_Name'] == cname:
                    out.setdefault(r['Kernel_Name'], []).append(float(r['Counter_Value']))
        return out

    fetch, write = counter(fetch_dir, 'FETCH_SIZE'), counter(write_dir, 'WRITE_SIZE')
    stats = {}
    for f in glob.glob(os.path.join(trace_dir, '**', '*kernel_stats.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            stats[r['Name']] = (float(r['AverageNs']) / 1e3, int(r['Calls']))
    for rec in recs:
        key = rec['rocprof_name']
        f = [v for k, vs in fetch.items() if key in k for v in vs]
        w = [v for k, vs in write.items() if key in k for v in vs]
        if f and w:
            # FETCH_SIZE / WRITE_SIZE are reported in KB on this rocprofv3; reads doubled (guide, HBM section)
            rec['hbm_read_bytes_pmc'] = int(2 * np.median(f) * 1024)
            rec['hbm_write_bytes_pmc'] = int(np.median(w) * 1024)
            rec['traffic_over_algorithmic'] = round((rec['hbm_read_bytes_pmc'] + rec['hbm_write_bytes_pmc']) /
                                                    rec['algorithmic_bytes'], 3)
        st = [v for k, v in stats.items() if key in k]
        if st:
            rec['us_per_launch_rocprof_avg_over_kernel_name'] = round(float(np.mean([s[0] for s in st])), 2)
    out = {'source': 'tools/elementwise_bench.py on MI355X: HIP-event timing of %d back-to-back launches per kernel; '
                     'rocprofv3 --kernel-trace --stats and separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of '
                     '`tools/elementwise_bench.py --once` (kernels that share a name -- the two softmax_argmax forms -- share a '
                     'rocprof row)' % 200,
           'workload': '%d images of %dx%d, C = %d, U = %d' % (N, W, H, C, U),
           'peak': 'HBM3E 8 TB/s (6.29 TB/s measured float4 copy, MI355X_MICROARCH.md)', 'kernels': recs}
    path = os.path.join(ROOT, 'profiles', '%s_elementwise.json' % tag)
    json.dump(out, open(path, 'w'), indent=1)
    print('wrote', path)


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--once', action='store_true', help='one launch per kernel (under rocprofv3)')
    ap.add_argument('--iters', type=int, default=200)
    ap.add_argument('--merge', nargs=3, metavar=('FETCH_DIR', 'WRITE_DIR', 'TRACE_DIR'))
    ap.add_argument('--tag', default='r3')
    a = ap.parse_args()
    if a.merge:
        merge(a.tag, *a.merge)
    else:
        print(json.dumps(measure(1 if a.once else a.iters)))
