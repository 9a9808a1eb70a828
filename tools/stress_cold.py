#!/usr/bin/env python3
"""Run-to-run comparison on COLD memory: every iteration frees the caching allocator's blocks and builds its operands in
freshly mapped device memory (first touch: cold TLBs, slow first loads), which widens timing windows that warm
benchmark loops never open.  Small deep-layer shapes (the ones the unit tests use).  GPU box only."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from modular_semantic_segmentation_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--iters', type=int, default=200)
args = ap.parse_args()
rng = np.random.default_rng(0)
bad = 0
_junk = torch.randn(4096, 4096, device='cuda', dtype=torch.bfloat16) * 50


def dirty_lds():
    """Leave every CU's LDS full of unrelated finite numbers (a GEMM and a sort stage their tiles there): a kernel that
    reads LDS it never wrote then computes with them instead of with its own leftovers from the previous iteration."""
    global _junk
    _junk = (_junk @ _junk).clamp_(-50, 50)
    torch.sort(_junk.view(-1)[: 1 << 22].float())


CASES = [(2, 2, 3, 512, 512, 3), (2, 4, 6, 512, 512, 3), (2, 8, 12, 256, 256, 3), (2, 2, 3, 64, 512, 1), (2, 4, 6, 512, 64, 1),
         (2, 16, 24, 128, 128, 3), (2, 32, 48, 64, 64, 3)]
for (n, h, w, cin, cout, k) in CASES:
    xs = rng.integers(-2, 3, (n, h, w, cin)).astype(np.float32)
    ws = rng.integers(-1, 2, (k, k, cin, cout)).astype(np.float32)
    dys = rng.integers(-2, 3, (n, h, w, cout)).astype(np.float32)
    refs = rng.integers(-1, 3, (n, h, w, cin)).astype(np.float32)
    want = None
    for it in range(args.iters):
        torch.cuda.empty_cache()
        pad = torch.empty(int(rng.integers(1, 64)) * 1024 * 1024, dtype=torch.uint8, device='cuda')   # shift the mappings
        x = ops.Act.from_dense(torch.from_numpy(xs).cuda())
        wt = torch.from_numpy(ws).cuda()
        wp = ops.pack_conv_weights(wt)
        wd = ops.pack_conv_weights_dgrad(wt)
        b = torch.zeros(cout, device='cuda')
        zb = torch.zeros(cin, device='cuda')
        dy = ops.Act.from_dense(torch.from_numpy(dys).cuda())
        ref = ops.Act.from_dense(torch.from_numpy(refs).cuda())
        dirty_lds()
        y, _ = ops.conv2d_fwd(x, wp, b, k, relu=True)
        dirty_lds()
        dx = ops.conv2d_bwd_data(dy, wd, zb, ops.Act(n, h, w, cin), k, relu_ref=ref)
        dirty_lds()
        dw = torch.zeros((k, k, cin, cout), device='cuda')
        db = torch.zeros(cout, device='cuda')
        ops.conv2d_bwd_filter(x, dy, dw, db, k)
        torch.cuda.synchronize()
        got = (y.t.clone(), dx.t.clone(), dw.clone(), db.clone())
        del pad
        if want is None:
            want = [g.cpu() for g in got]
            # the integer operands make every result exact: check the first one against torch on the CPU
            import torch.nn.functional as F
            y32 = F.conv2d(torch.from_numpy(xs).permute(0, 3, 1, 2), torch.from_numpy(ws).permute(3, 2, 0, 1), padding=(k - 1) // 2)
            assert torch.equal(y.interior().cpu(), torch.relu(y32).permute(0, 2, 3, 1).to(torch.bfloat16)), 'forward wrong'
        else:
            for name, g, wv in zip(('y', 'dx', 'dw', 'db'), got, want):
                if not torch.equal(g.cpu(), wv):
                    bad += 1
                    d = (g.cpu().float() - wv.float()).abs()
                    print('MISMATCH', (n, h, w, cin, cout, k), name, 'iter', it, 'n_diff', int((d > 0).sum()), 'max', float(d.max()), flush=True)
    print('case', (n, h, w, cin, cout, k), 'done', flush=True)

# ---- generation 4 (conv_f8_dma.hip: all operands by LDS-DMA, counted vmcnt across the item barrier): maps that tile in
# 16x32, bf16 (configuration 26) and e4m3 (24; 64- and 128-channel inputs: one and two chunks per tile), pooled outputs
G4 = [(2, 32, 64, 256, 128, 26), (3, 48, 96, 64, 64, 26), (2, 32, 64, 128, 128, 24), (3, 32, 96, 64, 128, 24)]
for (n, h, w, cin, cout, cfg) in G4:
    xs = rng.integers(-2, 3, (n, h, w, cin)).astype(np.float32)
    ws = rng.integers(-1, 2, (3, 3, cin, cout)).astype(np.float32)
    want = None
    for it in range(args.iters):
        torch.cuda.empty_cache()
        pad = torch.empty(int(rng.integers(1, 64)) * 1024 * 1024, dtype=torch.uint8, device='cuda')
        wt = torch.from_numpy(ws).cuda()
        b = torch.zeros(cout, device='cuda')
        if cfg == 24:
            x = ops.Act.from_dense(torch.from_numpy(xs).cuda(), dtype='fp8', scale_exp=0)
            wp, _ = ops.pack_conv_weights_f8(wt, scale_exp=0)
            y = ops.Act(n, h, w, cout, dtype='fp8', scale_exp=4)
            q = ops.Act(n, h // 2, w // 2, cout, dtype='fp8', scale_exp=4)
        else:
            x = ops.Act.from_dense(torch.from_numpy(xs).cuda())
            wp = ops.pack_conv_weights(wt)
            y, q = ops.Act(n, h, w, cout), ops.Act(n, h // 2, w // 2, cout)
        dirty_lds()
        ops.conv2d_fwd(x, wp, b, 3, relu=True, y=y, pooled=q, cfg=cfg)
        torch.cuda.synchronize()
        got = (y.t.view(torch.uint8).clone().cpu(), q.t.view(torch.uint8).clone().cpu())
        del pad
        if want is None:
            want = got
            if cfg == 26:
                import torch.nn.functional as F
                y32 = F.conv2d(torch.from_numpy(xs).permute(0, 3, 1, 2), torch.from_numpy(ws).permute(3, 2, 0, 1), padding=1)
                assert torch.equal(y.interior().cpu(), torch.relu(y32).permute(0, 2, 3, 1).to(torch.bfloat16)), 'forward wrong'
        else:
            for name, g, wv in zip(('y', 'pooled'), got, want):
                if not torch.equal(g, wv):
                    bad += 1
                    print('MISMATCH', (n, h, w, cin, cout, cfg), name, 'iter', it, 'n_diff', int((g != wv).sum()), flush=True)
    print('generation-4 case', (n, h, w, cin, cout, cfg), 'done', flush=True)

# ---- generation 5 (conv_col_dma.hip, configurations 27 / 28) and the two-model launches of generations 4 / 5
# (xv_conv2d_fwd_pair): the same conditions.  (Round 4: a compiler-promoted alloca in static LDS sat on top of these kernels'
# first patch buffer -- run-to-run differences that a loop over identical launches cannot see, because the leftovers in LDS
# are then the right values.)
import torch.nn.functional as F  # noqa: E402
G5 = [(2, 24, 48, 128, 64, 27, False), (3, 48, 16, 64, 128, 27, False), (2, 32, 32, 64, 128, 28, True),
      (2, 32, 64, 128, 128, 'pair', True), (2, 24, 48, 128, 64, 'pair', False), (9, 16, 32, 64, 64, 'pair', True)]
for (n, h, w, cin, cout, cfg, pool) in G5:
    ne = 2 if cfg == 'pair' else 1
    xs = [rng.integers(-2, 3, (n, h, w, cin)).astype(np.float32) for _ in range(ne)]
    ws = [rng.integers(-1, 2, (3, 3, cin, cout)).astype(np.float32) for _ in range(ne)]
    bs = [rng.integers(-3, 4, cout).astype(np.float32) for _ in range(ne)]
    want = None
    for it in range(args.iters):
        torch.cuda.empty_cache()
        pad = torch.empty(int(rng.integers(1, 64)) * 1024 * 1024, dtype=torch.uint8, device='cuda')
        x = [ops.Act.from_dense(torch.from_numpy(v).cuda()) for v in xs]
        wp = [ops.pack_conv_weights(torch.from_numpy(v).cuda()) for v in ws]
        b = [torch.from_numpy(v).cuda() for v in bs]
        y = [ops.Act(n, h, w, cout) for _ in range(ne)]
        q = [ops.Act(n, h // 2, w // 2, cout) if pool else None for _ in range(ne)]
        dirty_lds()
        if cfg == 'pair':
            assert ops.conv2d_fwd_pair(x[0], wp[0], b[0], x[1], wp[1], b[1], relu=True, ya=y[0], yb=y[1], pa=q[0], pb=q[1])
        else:
            ops.conv2d_fwd(x[0], wp[0], b[0], 3, relu=True, y=y[0], pooled=q[0], cfg=cfg)
        torch.cuda.synchronize()
        got = [t.t.clone().cpu() for t in y] + [t.t.clone().cpu() for t in q if t is not None]
        del pad
        if want is None:
            want = got
            for e in range(ne):
                y32 = F.conv2d(torch.from_numpy(xs[e]).permute(0, 3, 1, 2), torch.from_numpy(ws[e]).permute(3, 2, 0, 1),
                               torch.from_numpy(bs[e]), padding=1)
                assert torch.equal(y[e].interior().cpu(), torch.relu(y32).permute(0, 2, 3, 1).to(torch.bfloat16)), 'forward wrong'
        else:
            for i, (g, wv) in enumerate(zip(got, want)):
                if not torch.equal(g, wv):
                    bad += 1
                    print('MISMATCH', (n, h, w, cin, cout, cfg), 'output', i, 'iter', it, 'n_diff', int((g != wv).sum()), flush=True)
    print('generation-5 / pair case', (n, h, w, cin, cout, cfg), 'done', flush=True)
print('mismatches:', bad)
sys.exit(1 if bad else 0)
