#!/usr/bin/env python3
"""Run ONE conv layer shape with ONE tile configuration a few times (for rocprofv3 --pmc runs)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from modular_semantic_segmentation_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--scale', type=int, default=8)
ap.add_argument('--cin', type=int, default=512)
ap.add_argument('--cout', type=int, default=512)
ap.add_argument('--k', type=int, default=3)
ap.add_argument('--cfg', type=int, default=4)
ap.add_argument('--batch', type=int, default=8)
ap.add_argument('--iters', type=int, default=3)
ap.add_argument('--pool', action='store_true')
args = ap.parse_args()
h, w = 384 // args.scale, 768 // args.scale
x = ops.Act(args.batch, h, w, args.cin)
x.interior().normal_()
wt = torch.randn(args.k, args.k, args.cin, args.cout, device='cuda') * (1.0 / (args.k * args.k * args.cin) ** 0.5)
wp = ops.pack_conv_weights(wt)
b = torch.zeros(args.cout, device='cuda')
y = ops.Act(args.batch, h, w, args.cout)
q = ops.Act(args.batch, h // 2, w // 2, args.cout) if args.pool else None
for _ in range(args.iters):
    ops.conv2d_fwd(x, wp, b, args.k, y=y, pooled=q, cfg=args.cfg)
torch.cuda.synchronize()
print('done')
