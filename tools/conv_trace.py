#!/usr/bin/env python3
"""Phase breakdown of the generation-2 conv kernel from in-kernel cycle stamps (debug build only:
make -C modular_semantic_segmentation_amd/csrc clean all CXXFLAGS+=-DXV_CONV_TRACE).  For the first 32 work items of
every wave of every 32nd workgroup: [0] arrival at the item barrier, [1] barrier passed, [2] first fragments
requested, [3] MFMAs issued."""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from modular_semantic_segmentation_amd import _lib, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--scale', type=int, default=4)
ap.add_argument('--cin', type=int, default=256)
ap.add_argument('--cout', type=int, default=256)
ap.add_argument('--batch', type=int, default=16)
ap.add_argument('--pool', action='store_true')
ap.add_argument('--block', type=int, default=0)
args = ap.parse_args()
h, w = 384 // args.scale, 768 // args.scale
x = ops.Act(args.batch, h, w, args.cin)
x.interior().normal_()
wt = torch.randn(3, 3, args.cin, args.cout, device='cuda') * (1.0 / (9 * args.cin) ** 0.5)
wp = ops.pack_conv_weights(wt)
b = torch.zeros(args.cout, device='cuda')
y = ops.Act(args.batch, h, w, args.cout)
q = ops.Act(args.batch, h // 2, w // 2, args.cout) if args.pool else None
for _ in range(3):
    ops.conv2d_fwd(x, wp, b, 3, y=None if args.pool else y, pooled=q, write_y=not args.pool, cfg=17)
torch.cuda.synchronize()
buf = np.zeros((8, 8, 20, 6), dtype=np.int64)          # [traced block][wave][item][stamp]
fn = _lib.lib().xv_debug_read_trace
fn.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert fn(buf.ctypes.data, buf.nbytes) == 0
nchunks = args.cin // 32
t = buf[args.block]
t0 = t[:, :, 1].max(axis=0)                             # barrier release of each item = latest 'passed' stamp
print('traced workgroup %d; per item: cycles since the previous release, then per wave '
      '[arrive-before-release | release->first MFMA | taps | pack phase of the PREVIOUS item]' % (args.block * 32))
for it in range(2, 19):
    row = 'item %2d%s +%5d :' % (it, '*' if it % nchunks == nchunks - 1 else ' ', t0[it] - t0[it - 1])
    for wv in range(8):
        row += ' [%5d|%4d|%4d|%4d]' % (t0[it] - t[wv, it, 0], t[wv, it, 2] - t[wv, it, 1], t[wv, it, 3] - t[wv, it, 2],
                                        max(t[wv, it - 1, 4] - t[wv, it - 1, 3], 0))
    print(row)
