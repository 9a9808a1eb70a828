#!/usr/bin/env python3
"""Experiment (GPU box): conv1_1 -> conv1_2 (+pool) over the whole batch against the same two layers run strip by strip
(sub-batches whose conv1_1 output stays inside the 256 MiB Infinity Cache, ring of two strip buffers)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from modular_semantic_segmentation_amd import ops  # noqa: E402

N, H, W = 16, 384, 768
torch.manual_seed(0)
for cin in (3, 1):
    x = torch.rand(N, H, W, cin, device='cuda') * 255
    w1 = torch.randn(3, 3, cin, 64, device='cuda') * 0.05
    b1 = torch.zeros(64, device='cuda')
    w2 = ops.pack_conv_weights(torch.randn(3, 3, 64, 64, device='cuda') * 0.04)
    b2 = torch.zeros(64, device='cuda')
    pool = ops.Act(N, H // 2, W // 2, 64)

    class View(object):
        """a sub-batch view of an Act"""
        def __init__(self, act, b, e):
            self.n, self.h, self.w, self.c, self.dtype, self.scale_exp = e - b, act.h, act.w, act.c, act.dtype, act.scale_exp
            self.t = act.t[b:e]
            from modular_semantic_segmentation_amd._lib import xv_act
            self._xv = xv_act(self.t.data_ptr(), self.n, self.h, self.w, self.c, 0, 0)

        def xv(self):
            import ctypes
            return ctypes.byref(self._xv)

    def full():
        y1 = full.y1
        ops.conv2d_first_fwd(x, w1, b1, y1)
        ops.conv2d_fwd(y1, w2, b2, 3, pooled=pool, write_y=False)
    full.y1 = ops.Act(N, H, W, 64)

    def strips(sb, ring):
        bufs = strips.bufs[(sb, ring)]
        for k, b in enumerate(range(0, N, sb)):
            y1 = bufs[k % ring]
            ops.conv2d_first_fwd(x[b:b + sb], w1, b1, y1)
            ops.conv2d_fwd(y1, w2, b2, 3, pooled=View(pool, b, b + sb), write_y=False)
    strips.bufs = {}
    for sb in (1, 2, 4, 8):
        for ring in (1, 2):
            strips.bufs[(sb, ring)] = [ops.Act(sb, H, W, 64) for _ in range(ring)]

    def timeit(fn, *a):
        for _ in range(3):
            fn(*a)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            fn(*a)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 30 * 1e3
    for rep in range(3):
        print('cin %d: full batch %.1f us' % (cin, timeit(full)))
        ref = pool.t.clone()
        for sb, ring in ((4, 1), (4, 2), (8, 1), (2, 2)):
            t = timeit(strips, sb, ring)
            print('  strips of %d images, ring %d: %.1f us   same result: %s' % (sb, ring, t, torch.equal(pool.t, ref)))
