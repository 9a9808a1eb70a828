#!/usr/bin/env python3
"""Time the first conv layer (fp32 image in, 64-channel bf16 padded-NHWC out) at the bench shape (GPU box only)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from modular_semantic_segmentation_amd import ops  # noqa: E402

n, h, w = 8, 384, 768
for cin in (1, 3):
    x = torch.rand(n, h, w, cin, device='cuda') * 255
    wt = torch.randn(3, 3, cin, 64, device='cuda') * 0.05
    b = torch.randn(64, device='cuda')
    y = ops.Act(n, h, w, 64)
    for _ in range(3):
        ops.conv2d_first_fwd(x, wt, b, y, relu=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.conv2d_first_fwd(x, wt, b, y, relu=True)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print('cin %d: %.1f us  (%.2f TB/s of output)' % (cin, us, y.t.numel() * 2 / us / 1e6), flush=True)

# the write floor: a plain fill of the same output buffer
y = ops.Act(n, h, w, 64)
for _ in range(3):
    y.t.zero_()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    y.t.zero_()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
print('fill of the output buffer: %.1f us  (%.2f TB/s)' % (us, y.t.numel() * 2 / us / 1e6), flush=True)
