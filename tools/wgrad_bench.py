#!/usr/bin/env python3
"""The 3x3 filter gradient (xv_conv2d_bwd_filter_ws) on the FCN's layer shapes: microseconds and TFLOP/s per launch (GPU box).
  python tools/wgrad_bench.py [--batch 16] [--layers conv3_2,conv4_2]
XV_LIB=<path> (with XV_ALLOW_STALE_LIB=1): another build of the library, for A/B on one box (tools/wgrad_exp.sh)."""
import argparse
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from modular_semantic_segmentation_amd import _lib  # noqa: E402
if os.environ.get('XV_LIB'):
    _lib.LIB_PATH = os.environ['XV_LIB']
from modular_semantic_segmentation_amd import ops  # noqa: E402

SHAPES = {'conv1_2': (384, 768, 64, 64), 'conv2_1': (192, 384, 64, 128), 'conv2_2': (192, 384, 128, 128),
          'conv3_1': (96, 192, 128, 256), 'conv3_2': (96, 192, 256, 256), 'conv4_1': (48, 96, 256, 512),
          'conv4_2': (48, 96, 512, 512), 'conv5_1': (24, 48, 512, 512)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--layers', default=','.join(SHAPES))
    ap.add_argument('--variant', type=int, default=0, help='xv_set_wgrad_variant: 2 = round 5, 3 = loader waves (0: default)')
    ap.add_argument('--seconds', type=float, default=0.3, help='back-to-back launches before the timed ones (clock settles)')
    args = ap.parse_args()
    torch.manual_seed(0)
    assert _lib.lib().xv_set_wgrad_variant(args.variant) == 0
    for name in args.layers.split(','):
        h, w, cin, cout = SHAPES[name]
        n = args.batch
        x = ops.Act.from_dense(torch.relu(torch.randn(n, h, w, cin, device='cuda')))
        dy = ops.Act.from_dense(torch.randn(n, h, w, cout, device='cuda') * 1e-2)
        dw = torch.zeros(3, 3, cin, cout, device='cuda')
        db = torch.zeros(cout, device='cuda')
        ws = torch.empty(ops.conv2d_bwd_filter_workspace_bytes(x, cout, 3) // 4, device='cuda')

        def run():
            ops.conv2d_bwd_filter(x, dy, dw, db, 3, workspace=ws)
        import time
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < args.seconds:
            for _ in range(20):
                run()
            torch.cuda.synchronize()
        best = None
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                run()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 50 * 1e3
            best = us if best is None else min(best, us)
        fl = 2.0 * n * h * w * cin * cout * 9
        print('%-8s %4d images  %8.1f us  %7.0f TFLOP/s (incl. the slab reduce)' % (name, n, best, fl / best / 1e6), flush=True)


if __name__ == '__main__':
    main()
