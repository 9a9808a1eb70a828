#!/usr/bin/env python3
"""Time every conv tile configuration on every SimpleFCN layer shape (GPU box only).
Prints TFLOP/s per (layer, cfg); used to set the default choice in csrc/conv_mfma.hip."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from modular_semantic_segmentation_amd import _lib  # noqa: E402
if os.environ.get('XV_LIB'):            # another build of the library (A/B on one box; with XV_ALLOW_STALE_LIB=1)
    _lib.LIB_PATH = os.environ['XV_LIB']
from modular_semantic_segmentation_amd import ops  # noqa: E402

LAYERS = [('conv1_2', 1, 64, 64, 3, True), ('conv2_1', 2, 64, 128, 3, False), ('conv2_2', 2, 128, 128, 3, True),
          ('conv3_1', 4, 128, 256, 3, False), ('conv3_2', 4, 256, 256, 3, False), ('conv3_3', 4, 256, 256, 3, True),
          ('conv4_1', 8, 256, 512, 3, False), ('conv4_2', 8, 512, 512, 3, False), ('conv4_3', 8, 512, 512, 3, True),
          ('conv5_1', 16, 512, 512, 3, False), ('score_conv4', 8, 512, 64, 1, False),
          ('score_conv5', 16, 512, 64, 1, False)]


def tune_fp8(args):
    cfgs = [int(c) for c in args.cfgs.split(',')] if args.cfgs else [14, 15, 16, 24]
    print('layer        ' + ''.join('cfg%-6d' % c for c in cfgs) + ' default   (fp8 e4m3 operands, TFLOP/s)')
    for name, s, cin, cout, k, pool in LAYERS:
        if cin < 128:
            continue
        h, w = args.height // s, args.width // s
        x = ops.Act.from_dense(torch.randn(args.batch, h, w, cin, device='cuda').abs() * 40, dtype='fp8', scale_exp=0)
        wt = torch.randn(k, k, cin, cout, device='cuda') * (1.0 / (k * k * cin) ** 0.5)
        if args.data == 'zero':
            x = ops.Act(args.batch, h, w, cin, dtype='fp8', scale_exp=0)
            wt.zero_()
        wp, _ = ops.pack_conv_weights_f8(wt, scale_exp=0 if args.data == 'zero' else None)
        b = torch.zeros(cout, device='cuda')
        y = ops.Act(args.batch, h, w, cout, dtype='fp8', scale_exp=1)
        q = ops.Act(args.batch, h // 2, w // 2, cout, dtype='fp8', scale_exp=1) if pool else None
        flops = 2.0 * args.batch * h * w * cin * cout * k * k
        row = '%-12s ' % name
        yy = None if (pool and args.pooled_only) else y
        for cfg in cfgs + [-1]:
            try:
                for _ in range(2):
                    ops.conv2d_fwd(x, wp, b, k, y=yy, pooled=q, write_y=yy is not None, cfg=cfg)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.iters):
                    ops.conv2d_fwd(x, wp, b, k, y=yy, pooled=q, write_y=yy is not None, cfg=cfg)
                e1.record()
                torch.cuda.synchronize()
                row += '%-9.0f' % (flops / (e0.elapsed_time(e1) / args.iters) / 1e9)
            except _lib.XvError:
                row += '%-9s' % '-'
        print(row, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--height', type=int, default=384)
    ap.add_argument('--width', type=int, default=768)
    ap.add_argument('--iters', type=int, default=5)
    ap.add_argument('--cfgs', type=str, default='', help='comma-separated subset of configurations')
    ap.add_argument('--no-wgrad', action='store_true')
    ap.add_argument('--pooled-only', action='store_true', help='pool layers write only the pooled map')
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp8'], help="fp8: e4m3 operands and outputs (cin >= 128)")
    ap.add_argument('--data', default='normal', choices=['normal', 'zero', 'relu'],
                    help="operand values (bf16): 'zero' = all-zero maps and weights (no operand toggling: what the matrix pipes "
                         "reach when power does not hold the clock down), 'relu' = half the activations zero")
    args = ap.parse_args()
    if args.dtype == 'fp8':
        return tune_fp8(args)
    ncfg = _lib.lib().xv_conv2d_num_cfgs()
    cfgs = [int(c) for c in args.cfgs.split(',')] if args.cfgs else list(range(ncfg))
    print('layer        ' + ''.join('cfg%-6d' % c for c in cfgs) + ' default')
    for name, s, cin, cout, k, pool in LAYERS:
        h, w = args.height // s, args.width // s
        x = ops.Act(args.batch, h, w, cin)
        x.interior().normal_()
        wt = torch.randn(k, k, cin, cout, device='cuda') * (1.0 / (k * k * cin) ** 0.5)
        if args.data == 'zero':
            x.interior().zero_()
            wt.zero_()
        elif args.data == 'relu':
            x.interior().clamp_(min=0)
        wp = ops.pack_conv_weights(wt)
        b = torch.zeros(cout, device='cuda')
        y = ops.Act(args.batch, h, w, cout)
        q = ops.Act(args.batch, h // 2, w // 2, cout) if pool else None
        flops = 2.0 * args.batch * h * w * cin * cout * k * k
        row = '%-12s ' % name
        yy = None if (pool and args.pooled_only) else y
        for cfg in cfgs + [-1]:
            try:
                for _ in range(2):
                    ops.conv2d_fwd(x, wp, b, k, y=yy, pooled=q, write_y=yy is not None, cfg=cfg)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.iters):
                    ops.conv2d_fwd(x, wp, b, k, y=yy, pooled=q, write_y=yy is not None, cfg=cfg)
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / args.iters
                row += '%-9.0f' % (flops / ms / 1e9)
            except _lib.XvError:
                row += '%-9s' % '-'
        # filter gradient of the same layer (one configuration)
        dy = ops.Act(args.batch, h, w, cout)
        dy.interior().normal_()
        dw = torch.zeros((k, k, cin, cout), device='cuda')
        db = torch.zeros(cout, device='cuda')
        for variant in (() if args.no_wgrad else (1, 2)):
            _lib.lib().xv_set_wgrad_variant(variant)
            for wsp in (None, torch.empty(ops.conv2d_bwd_filter_workspace_bytes(x, cout, k) // 4, device='cuda')):
                for _ in range(2):
                    ops.conv2d_bwd_filter(x, dy, dw, db, k, workspace=wsp)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.iters):
                    ops.conv2d_bwd_filter(x, dy, dw, db, k, workspace=wsp)
                e1.record()
                torch.cuda.synchronize()
                row += ' wg%d%s %-5.0f' % (variant, 'a' if wsp is None else 's',
                                          flops / (e0.elapsed_time(e1) / args.iters) / 1e9)
        print(row, flush=True)


if __name__ == '__main__':
    main()
