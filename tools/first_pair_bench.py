#!/usr/bin/env python3
"""conv1_1 + conv1_2 (+ pool) fused (xv_conv_first_pair_fwd) against the two kernels, 16 images of 768x384 (GPU box)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from modular_semantic_segmentation_amd import _lib  # noqa: E402
if os.environ.get('XV_LIB'):            # A/B of library builds on one box (with XV_ALLOW_STALE_LIB=1)
    _lib.LIB_PATH = os.environ['XV_LIB']
from modular_semantic_segmentation_amd import ops  # noqa: E402

N, H, W = 16, 384, 768
torch.manual_seed(0)
for cin in (3, 1):
    x = torch.rand(N, H, W, cin, device='cuda') * (255 if cin == 3 else 65535)
    w1 = torch.randn(3, 3, cin, 64, device='cuda') * (0.02 if cin == 3 else 0.02 / 256)
    b1 = torch.zeros(64, device='cuda')
    w2 = ops.pack_conv_weights(torch.randn(3, 3, 64, 64, device='cuda') * 0.04)
    b2 = torch.zeros(64, device='cuda')
    pool, pool2, y1 = ops.Act(N, H // 2, W // 2, 64), ops.Act(N, H // 2, W // 2, 64), ops.Act(N, H, W, 64)

    def two():
        ops.conv2d_first_fwd(x, w1, b1, y1)
        ops.conv2d_fwd(y1, w2, b2, 3, pooled=pool, write_y=False)

    def fused():
        assert ops.conv_first_pair_fwd(x, w1, b1, w2, b2, pooled=pool2)

    def timeit(fn):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 50 * 1e3
    for rep in range(3):
        print('cin %d: two kernels %.1f us, fused %.1f us' % (cin, timeit(two), timeit(fused)), flush=True)
    print('  same pooled map:', torch.equal(pool.t, pool2.t))
