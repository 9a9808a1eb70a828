#!/usr/bin/env python3
"""Run-to-run comparison of the 3x3 filter gradient (xv_conv2d_bwd_filter_ws: slabs added in a fixed order -- the result must
be bitwise reproducible) on warm and on COLD memory, with foreign data left in every CU's LDS between launches: a race
between the loader waves' LDS-DMA and the compute waves' fragment reads in conv_wgrad_lw_kernel (one barrier per tile, in front of
its last halo row) would show as a rare mismatch.  Also: variant 3 against variant 2 on integer operands (equal sums).
GPU box only.   python tools/stress_wgrad.py [--iters 200]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from modular_semantic_segmentation_amd import _lib, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--iters', type=int, default=200)
args = ap.parse_args()
_junk = torch.randn(4096, 4096, device='cuda', dtype=torch.bfloat16) * 50


def dirty_lds():
    (_junk[:2048] @ _junk[:, :2048]).sum().item()
    torch.sort(torch.randn(1 << 20, device='cuda'))[0].sum().item()


bad = 0
SHAPES = [(16, 96, 192, 256, 256), (4, 48, 96, 512, 512), (16, 24, 48, 512, 512), (2, 40, 72, 128, 64), (1, 8, 32, 64, 64),
          (3, 17, 33, 64, 128), (8, 192, 384, 64, 64)]
for (n, h, w, cin, cout) in SHAPES:
    gen = torch.Generator(device='cuda').manual_seed(n * h + cin)
    xd = torch.relu(torch.randn(n, h, w, cin, device='cuda', generator=gen))
    dd = torch.randn(n, h, w, cout, device='cuda', generator=gen) * 1e-2
    first = None
    iters = max(8, args.iters // (1 + (n * h * w * cin * cout) // (1 << 28)))
    for it in range(iters):
        cold = it % 2 == 1
        if cold:
            torch.cuda.empty_cache()
            dirty_lds()
        x, dy = ops.Act.from_dense(xd), ops.Act.from_dense(dd)
        dw = torch.zeros(3, 3, cin, cout, device='cuda')
        db = torch.zeros(cout, device='cuda')
        ws = torch.full((ops.conv2d_bwd_filter_workspace_bytes(x, cout, 3) // 4,), float('nan'), device='cuda')
        ops.conv2d_bwd_filter(x, dy, dw, db, 3, workspace=ws)
        torch.cuda.synchronize()
        if first is None:
            first = (dw.clone(), db.clone())
            assert torch.isfinite(dw).all() and torch.isfinite(db).all()
        elif not (torch.equal(dw, first[0]) and torch.equal(db, first[1])):
            bad += 1
            print('MISMATCH', (n, h, w, cin, cout), 'iteration', it, 'cold' if cold else 'warm',
                  float((dw - first[0]).abs().max()), flush=True)
    # integer operands: variant 3 == variant 2 == exact
    xi = torch.randint(-2, 3, (n, h, w, cin), device='cuda', generator=gen).float()
    di = torch.randint(-1, 2, (n, h, w, cout), device='cuda', generator=gen).float()
    res = []
    for v in (2, 3):
        assert _lib.lib().xv_set_wgrad_variant(v) == 0
        x, dy = ops.Act.from_dense(xi), ops.Act.from_dense(di)
        dw = torch.zeros(3, 3, cin, cout, device='cuda')
        db = torch.zeros(cout, device='cuda')
        ws = torch.empty(ops.conv2d_bwd_filter_workspace_bytes(x, cout, 3) // 4, device='cuda')
        ops.conv2d_bwd_filter(x, dy, dw, db, 3, workspace=ws)
        torch.cuda.synchronize()
        res.append((dw, db))
    _lib.lib().xv_set_wgrad_variant(0)
    if not (torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])):
        bad += 1
        print('VARIANTS DIFFER on integers', (n, h, w, cin, cout), flush=True)
    print('shape', (n, h, w, cin, cout), iters, 'iterations ok' if bad == 0 else 'FAILED so far', flush=True)
print('mismatches:', bad)
sys.exit(1 if bad else 0)
