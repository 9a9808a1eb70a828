#!/usr/bin/env python3
"""conv1_1's filter gradient (xv_conv2d_first_bwd_filter_ws) at 16 x 768x384: microseconds per launch, RGB and depth (GPU box).
XV_LIB=<path> (with XV_ALLOW_STALE_LIB=1): another build of the library, for A/B on one box."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from modular_semantic_segmentation_amd import _lib  # noqa: E402
if os.environ.get('XV_LIB'):
    _lib.LIB_PATH = os.environ['XV_LIB']
from modular_semantic_segmentation_amd import ops  # noqa: E402

N, H, W = 16, 384, 768
torch.manual_seed(0)
for cin in (3, 1):
    x = torch.rand(N, H, W, cin, device='cuda') * (255 if cin == 3 else 65535)
    dy = ops.Act.from_dense(torch.randn(N, H, W, 64, device='cuda'))
    dw = torch.zeros(3, 3, cin, 64, device='cuda')
    db = torch.zeros(64, device='cuda')
    ws = torch.empty(ops.conv2d_first_bwd_filter_workspace_bytes(x) // 4, device='cuda')

    def run():
        ops.conv2d_first_bwd_filter(x, dy, dw, db, workspace=ws)
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        print('cin %d: %.1f us' % (cin, e0.elapsed_time(e1) / 20 * 1e3), flush=True)
