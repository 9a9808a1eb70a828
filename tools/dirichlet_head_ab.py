"""A/B of the fused Dirichlet head's forms (XV_DIRICHLET_HEAD_PK = 0: scalar, 1: packed fp32, 2: packed, two pixels per thread)
on the headline shape: labels must be equal, time per launch by HIP events."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
N, H, W, C, U = 16, 384, 768, 12, 64


def main():
    from modular_semantic_segmentation_amd import ops
    dev = torch.device('cuda:0')
    g = torch.Generator(device='cpu').manual_seed(0)
    hi, wi = H // 8, W // 8
    fa = ops.Act.from_dense(torch.rand((N, hi, wi, U), generator=g).to(dev))
    fb = ops.Act.from_dense(torch.rand((N, hi, wi, U), generator=g).to(dev))
    ws = torch.randn((U, C), generator=g).to(dev)
    bs = torch.randn(C, generator=g).to(dev)
    Sa = torch.zeros((N, hi + 2, wi + 2, C), device=dev)
    Sb = torch.zeros((N, hi + 2, wi + 2, C), device=dev)
    ops.score_lowres(fa, ws, C, Sa)
    ops.score_lowres(fb, ws * 0.7, C, Sb)
    am1 = torch.rand((2, C, C), generator=g).to(dev)
    lognorm = torch.randn((2, C), generator=g).to(dev)
    logprior = torch.randn(C, generator=g).to(dev)
    res, ref = {}, None
    for pk in (0, 1, 2, 4, 0, 1, 2, 4):
        os.environ['XV_DIRICHLET_HEAD_PK'] = str(pk)
        out = torch.empty((N, H, W), dtype=torch.int64, device=dev)
        run = lambda: ops.fused_head(Sa, Sb, bs, bs, N, hi, wi, C, am1, logprior, lognorm=lognorm, out=out)
        for _ in range(20):
            run()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(300):
            run()
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) / 300 * 1e3
        if ref is None:
            ref = out.clone()
        res.setdefault(str(pk), []).append({'us': round(us, 2), 'equal': bool(torch.equal(out, ref)),
                                            'classes_used': int(out.unique().numel())})
    print(json.dumps(res))


if __name__ == '__main__':
    main()
