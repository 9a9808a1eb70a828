"""A/B of the Dirichlet kernels' forms on the headline shape (XV_DIRICHLET_HEAD_PK / XV_DIRICHLET_FUSE_PK = 0: scalar, 1: packed
fp32): labels (and scores) must be equal, time per launch by HIP events."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
N, H, W, C, U = 16, 384, 768, int(sys.argv[1]) if len(sys.argv) > 1 else 12, 64   # argv[1]: another class count


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
    for pk in (0, 1, 0, 1):
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
    if C != 12:
        return
    # the unfused fusion kernel (probability maps -> fused label): XV_DIRICHLET_FUSE_PK = 0 scalar, P pixels per thread packed
    pa = torch.softmax(torch.randn((N, H, W, C), generator=g), -1).to(dev)
    pb = torch.softmax(torch.randn((N, H, W, C), generator=g), -1).to(dev)
    res, ref = {}, None
    for pk in (0, 1, 0, 1):
        os.environ['XV_DIRICHLET_FUSE_PK'] = str(pk)
        run = lambda: ops.dirichlet_fuse([pa, pb], am1, lognorm, logprior)
        for _ in range(20):
            run()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(300):
            run()
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) / 300 * 1e3
        lab, sc = ops.dirichlet_fuse([pa, pb], am1, lognorm, logprior, want_score=True)
        if ref is None:
            ref = (lab.clone(), sc.clone())
        res.setdefault(str(pk), []).append({'us': round(us, 2), 'frac_8TBps': round(N * H * W * 104 / us / 8e6, 3),
                                            'equal': bool(torch.equal(lab, ref[0]) and torch.equal(sc, ref[1]))})
    print(json.dumps(res))


if __name__ == '__main__':
    main()
