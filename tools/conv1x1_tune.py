#!/usr/bin/env python3
"""Time every first-generation tile configuration on the 1x1 conv shapes of the AdapNet expert at 768x384
(block stages, shortcuts and the im2col'ed atrous pairs): TFLOP/s per (shape, cfg)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from modular_semantic_segmentation_amd import _lib, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=16)
ap.add_argument('--iters', type=int, default=10)
args = ap.parse_args()
SHAPES = [(48, 96, 64, 512), (24, 48, 64, 512),    # the data gradients of the FCN's two score convs
          (96, 192, 64, 64), (96, 192, 64, 256), (96, 192, 256, 64), (48, 96, 256, 128), (48, 96, 128, 512),
          (48, 96, 512, 128), (48, 96, 2304, 64), (24, 48, 512, 256), (24, 48, 256, 1024), (24, 48, 1024, 256),
          (24, 48, 4608, 256), (24, 48, 1024, 512), (24, 48, 9216, 512), (24, 48, 512, 2048), (24, 48, 2048, 512),
          (24, 48, 1024, 2048), (24, 48, 2048, 64)]
ncfg = _lib.lib().xv_conv2d_num_cfgs()
print('%-24s' % 'h x w x cin -> cout' + ''.join('%6d' % c for c in range(ncfg)) + '   best')
for h, w, cin, cout in SHAPES:
    x = ops.Act(args.batch, h, w, cin)
    x.interior().normal_()
    wp = ops.pack_conv_weights(torch.randn(1, 1, cin, cout, device='cuda') * cin ** -0.5)
    b = torch.zeros(cout, device='cuda')
    y = ops.Act(args.batch, h, w, cout)
    flops = 2.0 * args.batch * h * w * cin * cout
    row, best = [], (0.0, -1)
    for cfg in range(ncfg):                         # configurations that cannot run a shape report XV_ESHAPE
        try:
            ops.conv2d_fwd(x, wp, b, 1, y=y, cfg=cfg)
        except _lib.XvError:
            row.append('     -')
            continue
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            ops.conv2d_fwd(x, wp, b, 1, y=y, cfg=cfg)
        e1.record()
        torch.cuda.synchronize()
        tf = flops * args.iters / (e0.elapsed_time(e1) * 1e-3) / 1e12
        row.append('%6.0f' % tf)
        best = max(best, (tf, cfg))
    # what the chooser picks (pick_cfg, conv_mfma.hip)
    ops.conv2d_fwd(x, wp, b, 1, y=y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        ops.conv2d_fwd(x, wp, b, 1, y=y)
    e1.record()
    torch.cuda.synchronize()
    auto = flops * args.iters / (e0.elapsed_time(e1) * 1e-3) / 1e12
    print('%-24s' % ('%dx%dx%d->%d' % (h, w, cin, cout)) + ''.join(row) + '   cfg %d (%.0f)   chooser %.0f' % (best[1], best[0], auto))
