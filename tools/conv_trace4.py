#!/usr/bin/env python3
"""Phase breakdown of the generation-4 conv kernel (conv_dma4_kernel) from in-kernel cycle stamps (`make -C csrc trace`
-> tools/build/libxview_hip_trace.so).  For the first 24 work items of waves 0 and 4 of every 32nd workgroup:
[0] arrival at the item barrier, [1] barrier passed, [2] all MFMAs issued, [3] tile epilogue done."""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from modular_semantic_segmentation_amd import _lib  # noqa: E402
_lib.LIB_PATH = os.path.join(ROOT, 'tools', 'build', 'libxview_hip_trace.so')
from modular_semantic_segmentation_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--scale', type=int, default=1)
ap.add_argument('--cin', type=int, default=64)
ap.add_argument('--cout', type=int, default=64)
ap.add_argument('--batch', type=int, default=16)
ap.add_argument('--mode', default='pool', choices=['y', 'pool', 'both'])
ap.add_argument('--cfg', type=int, default=26)
ap.add_argument('--block', type=int, default=0)
ap.add_argument('--data', default='normal')
ap.add_argument('--dgrad', action='store_true', help='the data-gradient kernel: addend + relu mask epilogue (mode y)')
args = ap.parse_args()
h, w = 384 // args.scale, 768 // args.scale
x = ops.Act(args.batch, h, w, args.cin)
wt = torch.randn(3, 3, args.cin, args.cout, device='cuda') * (1.0 / (9 * args.cin) ** 0.5)
if args.data == 'zero':
    wt.zero_()
else:
    x.interior().normal_()
wp = ops.pack_conv_weights(wt)
b = torch.zeros(args.cout, device='cuda')
y = ops.Act(args.batch, h, w, args.cout) if args.mode != 'pool' else None
q = ops.Act(args.batch, h // 2, w // 2, args.cout) if args.mode != 'y' else None
if args.dgrad:
    wd = ops.pack_conv_weights_dgrad(torch.randn(3, 3, args.cout, args.cin, device='cuda') * (1.0 / (9 * args.cin) ** 0.5))
    ref, add, dx = ops.Act(args.batch, h, w, args.cout), ops.Act(args.batch, h, w, args.cout), ops.Act(args.batch, h, w, args.cout)
    ref.interior().normal_()
    add.interior().normal_()
    for _ in range(20):
        ops.conv2d_bwd_data(x, wd, b, dx, 3, relu_ref=ref, addend=add)
else:
  for _ in range(20):
    ops.conv2d_fwd(x, wp, b, 3, y=y, pooled=q, write_y=y is not None, cfg=args.cfg)
torch.cuda.synchronize()
buf = np.zeros((8, 2, 24, 4), dtype=np.int64)          # [traced block][wave 0 / wave 4: the two waves of one SIMD][item][stamp]
fn = _lib.lib().xv_debug_read_trace4
fn.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert fn(buf.ctypes.data, buf.nbytes) == 0
nchunks = args.cin // 32
t = buf[args.block]
rel = t[:, :, 1].max(axis=0)                            # barrier release of each item = latest 'passed' stamp
print('cfg %d, %dx%d x %d, %d -> %d channels, mode %s; traced workgroup %d' % (args.cfg, w, h, args.batch, args.cin, args.cout,
                                                                                args.mode, args.block * 32))
print('per item: cycles since the previous release; per wave [wait at the barrier | release -> MFMAs issued | epilogue]')
tot = np.zeros(3)
for it in range(2, 22):
    row = 'item %2d%s +%5d :' % (it, '*' if it % nchunks == nchunks - 1 else ' ', rel[it] - rel[it - 1])
    for wv in range(2):
        a, bq, c = rel[it] - t[wv, it, 0], t[wv, it, 2] - t[wv, it, 1], t[wv, it, 3] - t[wv, it, 2]
        tot += (a, bq, c)
        row += ' [%5d|%5d|%5d]' % (a, bq, c)
    print(row)
n = 20 * 2
print('mean per item and wave: barrier wait %.0f, taps %.0f, epilogue %.0f; item period %.0f cycles' % (
    tot[0] / n, tot[1] / n, tot[2] / n, (rel[21] - rel[1]) / 20.0))
