#!/usr/bin/env python3
"""Run the same conv launch / the same model prediction many times and compare bit for bit (GPU box only): a race in
the hand-synchronised kernels (counted vmcnt / lgkmcnt, LDS-DMA double buffers) would show as a rare mismatch."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from modular_semantic_segmentation_amd import get_model, ops  # noqa: E402

torch.manual_seed(0)
bad = 0
for (n, h, w, cin, cout, pool) in [(16, 48, 96, 512, 512, False), (16, 96, 192, 256, 256, True), (8, 192, 384, 64, 128, False),
                                   (4, 384, 768, 64, 64, True), (3, 40, 72, 128, 64, False), (16, 24, 48, 512, 512, False)]:
    x = ops.Act(n, h, w, cin)
    x.interior().normal_()
    wt = torch.randn(3, 3, cin, cout, device='cuda') * (1.0 / (9 * cin) ** 0.5)
    wp = ops.pack_conv_weights(wt)
    b = torch.randn(cout, device='cuda')
    ref_y = ref_q = None
    for it in range(int(os.environ.get("XV_STRESS_ITERS", "150"))):
        y = ops.Act(n, h, w, cout)
        q = ops.Act(n, h // 2, w // 2, cout) if pool else None
        ops.conv2d_fwd(x, wp, b, 3, y=y, pooled=q)
        if ref_y is None:
            ref_y, ref_q = y.t.clone(), (q.t.clone() if pool else None)
        else:
            if not torch.equal(y.t, ref_y) or (pool and not torch.equal(q.t, ref_q)):
                bad += 1
                d = (y.t.float() - ref_y.float()).abs()
                print('MISMATCH conv', (n, h, w, cin, cout, pool), 'iter', it, 'n_diff', int((d > 0).sum()), 'max', float(d.max()))
    torch.cuda.synchronize()
    print('conv', (n, h, w, cin, cout, pool), 'done')

# generation 2b (item barrier inside tap 8), the 24x16 tile of generation 2 and the fp8 kernel: the same run-to-run comparison
for (n, h, w, cin, cout, pool, kind) in [(16, 48, 96, 512, 512, False, 'cfg21'), (16, 96, 192, 256, 256, True, 'cfg21'),
                                         (4, 384, 768, 64, 64, True, 'cfg21'), (3, 40, 72, 128, 64, False, 'cfg21'),
                                         (16, 24, 48, 512, 512, False, 'cfg22'), (5, 30, 40, 128, 64, False, 'cfg22'),
                                         (4, 128, 256, 512, 512, True, 'fp8'), (8, 48, 96, 256, 128, False, 'fp8'),
                                         (3, 40, 72, 128, 64, False, 'fp8')]:
    wt = torch.randn(3, 3, cin, cout, device='cuda') * (1.0 / (9 * cin) ** 0.5)
    b = torch.randn(cout, device='cuda')
    if kind == 'fp8':
        x = ops.Act.from_dense(torch.randn(n, h, w, cin, device='cuda').abs() * 30, dtype='fp8', scale_exp=0)
        wp, _ = ops.pack_conv_weights_f8(wt)
        okw = dict(dtype='fp8', scale_exp=1)
        cfg = -1
    else:
        x = ops.Act(n, h, w, cin)
        x.interior().normal_()
        wp = ops.pack_conv_weights(wt)
        okw = {}
        cfg = int(kind[3:])
    ref_y = ref_q = None
    for it in range(int(os.environ.get("XV_STRESS_ITERS", "150"))):
        y = ops.Act(n, h, w, cout, **okw)
        q = ops.Act(n, h // 2, w // 2, cout, **okw) if pool else None
        ops.conv2d_fwd(x, wp, b, 3, y=y, pooled=q, cfg=cfg)
        raw_y, raw_q = y.t.view(torch.uint8), (q.t.view(torch.uint8) if pool else None)
        if ref_y is None:
            ref_y, ref_q = raw_y.clone(), (raw_q.clone() if pool else None)
        elif not torch.equal(raw_y, ref_y) or (pool and not torch.equal(raw_q, ref_q)):
            bad += 1
            print('MISMATCH', kind, (n, h, w, cin, cout, pool), 'iter', it)
    torch.cuda.synchronize()
    print(kind, (n, h, w, cin, cout, pool), 'done')

# 1x1 convs: the flat-GEMM kernel (double-buffered LDS-DMA, interior-predicated stores) and the 128-channel tile
for (n, h, w, cin, cout) in [(16, 24, 48, 4608, 256), (8, 24, 48, 1024, 2048), (5, 17, 23, 256, 128), (4, 48, 96, 128, 512)]:
    x = ops.Act(n, h, w, cin)
    x.interior().normal_()
    wp = ops.pack_conv_weights(torch.randn(1, 1, cin, cout, device='cuda') * cin ** -0.5)
    b = torch.randn(cout, device='cuda')
    res = ops.Act(n, h, w, cout)
    res.interior().normal_()
    ref_y = None
    for it in range(int(os.environ.get("XV_STRESS_ITERS", "150"))):
        y = ops.conv1x1_residual(x, wp, b, res, relu=True, y=ops.Act(n, h, w, cout))
        if ref_y is None:
            ref_y = y.t.clone()
        elif not torch.equal(y.t, ref_y):
            bad += 1
            print('MISMATCH conv1x1', (n, h, w, cin, cout), 'iter', it)
    torch.cuda.synchronize()
    print('conv1x1', (n, h, w, cin, cout), 'done')

g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'notebook_868.npz'))
desc = ({'rgb': 'float32', 'depth': 'float32', 'labels': 'int32'},
        {'rgb': (None, None, 3), 'depth': (None, None, 1), 'labels': (None, None)}, 12)
net = get_model('bayes_fusion')(data_description=desc, confusion_matrices={'rgb': g['cm_rgb'], 'depth': g['cm_depth']},
                                num_units=64, prefixes={'rgb': 'rgb', 'depth': 'depth'},
                                num_channels={'rgb': 3, 'depth': 1}, expert_model='fcn', batchsize=2, seed=4)
rng = np.random.default_rng(11)
a = {'rgb': rng.integers(0, 256, (2, 64, 96, 3)).astype(np.float32), 'depth': rng.integers(0, 65536, (2, 64, 96, 1)).astype(np.float32)}
b2 = {'rgb': rng.integers(0, 256, (2, 64, 96, 3)).astype(np.float32), 'depth': rng.integers(0, 65536, (2, 64, 96, 1)).astype(np.float32)}
ea, eb = net.predict(a), net.predict(b2)
for it in range(100):
    if not np.array_equal(net.predict(a), ea) or not np.array_equal(net.predict(b2), eb):
        bad += 1
        print('MISMATCH eager model iter', it)
net.capture_graph({k: torch.from_numpy(v).cuda() for k, v in a.items()})
for it in range(200):
    pa, pb = net.predict(a), net.predict(b2)
    if not np.array_equal(pa, ea) or not np.array_equal(pb, eb):
        bad += 1
        print('MISMATCH graph model iter', it, int((pa != ea).sum()), int((pb != eb).sum()))
print('mismatches:', bad)
sys.exit(1 if bad else 0)
