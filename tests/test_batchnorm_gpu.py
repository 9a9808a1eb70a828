"""Training-mode batch norm and the un-commuted training head (csrc/batchnorm.hip) against PyTorch-CPU autograd of the
same [TF1] formulas (tf.layers.batch_normalization(training=True): biased batch variance, eps 1e-3, momentum 0.99 with
the unbiased variance in the moving average)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import fcn_oracle as fo


@pytest.fixture(scope='module')
def ops():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from modular_semantic_segmentation_amd import ops as o
    return o


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _nchw(a):
    return torch.from_numpy(a).permute(0, 3, 1, 2).contiguous()


@pytest.mark.parametrize('c,relu', [(64, True), (128, False), (512, True)])
def test_bn_forward_backward(ops, c, relu):
    rng = np.random.default_rng(c)
    n, h, w = 2, 12, 20
    z = fo.round_bf16((rng.standard_normal((n, h, w, c)) * 1.5 + 0.3).astype(np.float32))
    gamma = rng.uniform(0.5, 1.5, c).astype(np.float32)
    beta = (0.3 * rng.standard_normal(c)).astype(np.float32)
    mm0 = (0.1 * rng.standard_normal(c)).astype(np.float32)
    mv0 = rng.uniform(0.5, 1.5, c).astype(np.float32)
    dy = fo.round_bf16(rng.standard_normal((n, h, w, c)).astype(np.float32))
    # reference: autograd through the explicit formula
    zt = _nchw(z).double().requires_grad_(True)
    g, b = torch.from_numpy(gamma).double().requires_grad_(True), torch.from_numpy(beta).double().requires_grad_(True)
    mean = zt.mean(dim=(0, 2, 3), keepdim=True)
    var = ((zt - mean) ** 2).mean(dim=(0, 2, 3), keepdim=True)
    yt = (zt - mean) / torch.sqrt(var + 1e-3) * g.view(1, -1, 1, 1) + b.view(1, -1, 1, 1)
    if relu:
        yt = F.relu(yt)
    yt.backward(_nchw(dy).double())
    # GPU
    za, ya = ops.Act.from_dense(_dev(z)), ops.Act(n, h, w, c)
    st = ops.BnState(c, 'cuda')
    mm, mv = _dev(mm0.copy()), _dev(mv0.copy())
    ops.bn_forward(za, _dev(gamma), _dev(beta), mm, mv, st, ya, relu=relu)
    dgamma, dbeta = torch.zeros(c, device='cuda'), torch.zeros(c, device='cuda')
    dza = ops.Act(n, h, w, c)
    ops.bn_backward(ops.Act.from_dense(_dev(dy)), ya if relu else None, za, _dev(gamma), st, dgamma, dbeta, dza)
    torch.cuda.synchronize()
    y_ref = yt.detach().permute(0, 2, 3, 1).numpy()
    got_y = ya.interior().float().cpu().numpy()
    np.testing.assert_allclose(got_y, y_ref, rtol=2 ** -7, atol=2e-3)
    M = n * h * w
    np.testing.assert_allclose(mm.cpu().numpy(), 0.99 * mm0 + 0.01 * mean.detach().numpy().ravel(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(mv.cpu().numpy(), 0.99 * mv0 + 0.01 * var.detach().numpy().ravel() * M / (M - 1),
                               rtol=1e-5, atol=1e-6)
    # the relu mask of the kernel is y > 0 on the ROUNDED output: compare where the reference is not at a rounding edge
    np.testing.assert_allclose(dgamma.cpu().numpy(), g.grad.numpy(), rtol=2e-2, atol=5e-2)
    np.testing.assert_allclose(dbeta.cpu().numpy(), b.grad.numpy(), rtol=2e-2, atol=5e-2)
    dz_ref = zt.grad.permute(0, 2, 3, 1).numpy()
    got = dza.interior().float().cpu().numpy()
    assert np.abs(got - dz_ref).max() < 2e-2 * np.abs(dz_ref).max() + 1e-3
    if relu:
        # the default recomputed the relu mask from z (z * scale + shift > 0); read from the activation map instead, the
        # gradients are the same bits (the reductions go through fp64 atomics: compare after rounding to fp32 precision)
        dg2, db2, dz2 = torch.zeros(c, device='cuda'), torch.zeros(c, device='cuda'), ops.Act(n, h, w, c)
        ops.bn_backward(ops.Act.from_dense(_dev(dy)), ya, za, _dev(gamma), st, dg2, db2, dz2, mask_from_z=False)
        torch.cuda.synchronize()
        np.testing.assert_allclose(dg2.cpu().numpy(), dgamma.cpu().numpy(), rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(db2.cpu().numpy(), dbeta.cpu().numpy(), rtol=1e-6, atol=1e-6)
        assert (dz2.t != dza.t).float().mean().item() < 1e-3


def test_bn_dense_forward_backward(ops):
    rng = np.random.default_rng(3)
    rows, c = 3000, 12
    z = rng.standard_normal((rows, c)).astype(np.float32) * 2 + 0.5
    gamma, beta = rng.uniform(0.5, 1.5, c).astype(np.float32), rng.standard_normal(c).astype(np.float32)
    dy = rng.standard_normal((rows, c)).astype(np.float32)
    zt = torch.from_numpy(z).double().requires_grad_(True)
    g, b = torch.from_numpy(gamma).double().requires_grad_(True), torch.from_numpy(beta).double().requires_grad_(True)
    mean, var = zt.mean(0), ((zt - zt.mean(0)) ** 2).mean(0)
    yt = (zt - mean) / torch.sqrt(var + 1e-3) * g + b
    yt.backward(torch.from_numpy(dy).double())
    st = ops.BnState(c, 'cuda')
    zd, y = _dev(z), torch.empty((rows, c), device='cuda')
    mm, mv = torch.zeros(c, device='cuda'), torch.ones(c, device='cuda')
    ops.bn_dense_forward(zd, _dev(gamma), _dev(beta), mm, mv, st, y)
    dgamma, dbeta, dz = torch.zeros(c, device='cuda'), torch.zeros(c, device='cuda'), torch.empty((rows, c), device='cuda')
    ops.bn_dense_backward(_dev(dy), zd, _dev(gamma), st, dgamma, dbeta, dz)
    torch.cuda.synchronize()
    np.testing.assert_allclose(y.cpu().numpy(), yt.detach().numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(dgamma.cpu().numpy(), g.grad.numpy(), rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(dbeta.cpu().numpy(), b.grad.numpy(), rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(dz.cpu().numpy(), zt.grad.numpy(), rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize('factor', [2, 8])
def test_upsample_raw_and_its_transpose(ops, factor):
    rng = np.random.default_rng(factor)
    n, h, w, c = 2, 5, 7, 64
    x = fo.round_bf16(rng.standard_normal((n, h, w, c)).astype(np.float32))
    xt = _nchw(x).requires_grad_(True)
    yt = fo.deconv_same(xt, fo.bilinear_kernel(2 * factor, c), factor, relu=False)
    y = ops.upsample_raw_fwd(ops.Act.from_dense(_dev(x)), factor)
    torch.cuda.synchronize()
    ref = yt.detach().permute(0, 2, 3, 1).numpy()
    np.testing.assert_allclose(y.interior().float().cpu().numpy(), ref, rtol=2 ** -7, atol=1e-5)
    dy = fo.round_bf16(rng.standard_normal((n, factor * h, factor * w, c)).astype(np.float32))
    yt.backward(_nchw(dy))
    dx = ops.upsample_raw_bwd(ops.Act.from_dense(_dev(dy)), factor, ops.Act(n, h, w, c))
    torch.cuda.synchronize()
    dref = xt.grad.permute(0, 2, 3, 1).numpy()
    got = dx.interior().float().cpu().numpy()
    assert np.abs(got - dref).max() < 2 ** -7 * np.abs(dref).max() + 1e-4
    if factor == 8:
        # the two forms of the x8 gradient (per-block sums in a workspace: the default; the 256-tap gather): the same sums in
        # two orders -- equal to the rounding of the bf16 result
        import modular_semantic_segmentation_amd.ops as o
        saved = o.UPS8_BLOCK_SUMS
        try:
            o.UPS8_BLOCK_SUMS = False
            dx2 = ops.upsample_raw_bwd(ops.Act.from_dense(_dev(dy)), factor, ops.Act(n, h, w, c))
        finally:
            o.UPS8_BLOCK_SUMS = saved
        torch.cuda.synchronize()
        assert torch.equal(dx.t[:, 0], dx2.t[:, 0]) and torch.equal(dx.t[:, :, 0], dx2.t[:, :, 0])        # zero border untouched
        assert (dx.t.float() - dx2.t.float()).abs().max().item() <= 2 ** -7 * np.abs(dref).max()


@pytest.mark.parametrize('C,shape', [(12, (2, 8, 12)), (5, (2, 8, 12)), (12, (2, 16, 64)), (16, (1, 8, 32)), (5, (3, 24, 96))])
def test_dense_score_conv_and_cross_entropy(ops, C, shape):
    """(maps that tile in 8x32 pixels take the score layer's filter gradient on the bf16 matrix instruction with an exact
    three-way split of the fp32 score gradient -- conv_wgrad.hip score_dense_wgrad_split_kernel -- the others the fp32 form)"""
    rng = np.random.default_rng(C)
    (n, h, w), U = shape, 64
    u = fo.round_bf16(np.abs(rng.standard_normal((n, h, w, U))).astype(np.float32))
    ws = (0.3 * rng.standard_normal((U, C))).astype(np.float32)
    bs = rng.standard_normal(C).astype(np.float32)
    labels = rng.integers(-1, C, (n, h, w)).astype(np.int32)
    ut = torch.from_numpy(u).requires_grad_(True)
    wt, bt = torch.from_numpy(ws).requires_grad_(True), torch.from_numpy(bs).requires_grad_(True)
    score_t = ut.reshape(-1, U) @ wt + bt
    lab = torch.from_numpy(labels.reshape(-1).astype(np.int64))
    valid = lab >= 0
    logp = F.log_softmax(score_t, dim=1)
    loss_t = -(logp[valid, lab[valid]]).sum() / (1e-20 + valid.sum())
    loss_t.backward()
    ua = ops.Act.from_dense(_dev(u))
    score = torch.empty((n, h, w, C), device='cuda')
    ops.score_dense_fwd(ua, _dev(ws), _dev(bs), C, score)
    count = torch.tensor([int(valid.sum())], dtype=torch.int64, device='cuda')
    loss = torch.zeros(1, dtype=torch.float64, device='cuda')
    dlogits = torch.empty_like(score)
    ops.softmax_ce_dense(score, _dev(labels), count, C, loss, dlogits)
    dws, dbs, du = torch.zeros((U, C), device='cuda'), torch.zeros(C, device='cuda'), ops.Act(n, h, w, U)
    ops.score_dense_bwd(ua, dlogits, _dev(ws), C, dws, dbs, du)
    torch.cuda.synchronize()
    np.testing.assert_allclose(score.cpu().numpy().reshape(-1, C), score_t.detach().numpy(), rtol=1e-4, atol=1e-4)
    assert abs(loss.item() - loss_t.item()) < 1e-5 * abs(loss_t.item()) + 1e-7
    np.testing.assert_allclose(dws.cpu().numpy(), wt.grad.numpy(), rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(dbs.cpu().numpy(), bt.grad.numpy(), rtol=1e-3, atol=1e-6)
    got = du.interior().float().cpu().numpy()
    ref = ut.grad.numpy()
    assert np.abs(got - ref).max() < 2 ** -7 * np.abs(ref).max() + 1e-7


@pytest.mark.parametrize('c', [64, 256])
def test_bn_pool_backward_equals_the_three_pass_form(ops, c):
    """conv -> batch norm -> relu -> 2x2 max-pool: the fused gradient (MaxPoolGrad + ReluGrad + batch-norm gradient in the
    two normalisation passes, from the gradient of the POOLED map) against xv_maxpool2x2_bwd + xv_bn_bwd on the same
    tensors: same routing (first positive maximum of the stored activations, ties included), same values up to the order
    of the per-channel sums."""
    rng = np.random.default_rng(c + 1)
    n, h, w = 2, 12, 20
    z = fo.round_bf16((rng.standard_normal((n, h, w, c)) * 1.5 + 0.3).astype(np.float32))
    z[:, ::4, ::4] = z[:, 1::4, 1::4]                                    # exact ties inside some windows
    gamma = rng.uniform(0.5, 1.5, c).astype(np.float32)
    beta = (0.3 * rng.standard_normal(c)).astype(np.float32)
    dp = fo.round_bf16(rng.standard_normal((n, h // 2, w // 2, c)).astype(np.float32))
    za, ya = ops.Act.from_dense(_dev(z)), ops.Act(n, h, w, c)
    st = ops.BnState(c, 'cuda')
    ops.bn_forward(za, _dev(gamma), _dev(beta), torch.zeros(c, device='cuda'), torch.ones(c, device='cuda'), st, ya, relu=True)
    dpa = ops.Act.from_dense(_dev(dp))
    routed = ops.maxpool2x2_bwd(ya, dpa, ops.Act(n, h, w, c))
    dg1, db1, dz1 = torch.zeros(c, device='cuda'), torch.zeros(c, device='cuda'), ops.Act(n, h, w, c)
    ops.bn_backward(routed, ya, za, _dev(gamma), st, dg1, db1, dz1)
    dg2, db2, dz2 = torch.zeros(c, device='cuda'), torch.zeros(c, device='cuda'), ops.Act(n, h, w, c)
    ops.bn_pool_backward(dpa, za, _dev(gamma), st, dg2, db2, dz2)
    torch.cuda.synchronize()
    # the per-channel sums are the same numbers added in another order (pooled-pixel walk against full-resolution walk):
    # equal to fp32 rounding; dz then differs only where that moves a value across a bf16 rounding boundary
    np.testing.assert_allclose(dg2.cpu().numpy(), dg1.cpu().numpy(), rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(db2.cpu().numpy(), db1.cpu().numpy(), rtol=2e-5, atol=2e-5)
    a, b = dz1.t.float(), dz2.t.float()
    assert (a != b).float().mean().item() < 2e-3
    assert ((a - b).abs() <= 2.0 ** -7 * a.abs() + 1e-6).all()


def test_bn_apply_with_fused_pool(ops):
    """xv_bn_apply_pool: the activation and its 2x2 max-pool from one pass equal xv_bn_apply followed by xv_maxpool2x2_fwd,
    bit for bit; without a full-resolution output only the pooled map is written."""
    rng = np.random.default_rng(3)
    n, h, w, c = 2, 12, 20, 128
    z = fo.round_bf16((rng.standard_normal((n, h, w, c)) * 1.5 + 0.3).astype(np.float32))
    gamma, beta = rng.uniform(0.5, 1.5, c).astype(np.float32), (0.3 * rng.standard_normal(c)).astype(np.float32)
    za = ops.Act.from_dense(_dev(z))
    st = ops.BnState(c, 'cuda')
    y1 = ops.Act(n, h, w, c)
    ops.bn_forward(za, _dev(gamma), _dev(beta), torch.zeros(c, device='cuda'), torch.ones(c, device='cuda'), st, y1, relu=True)
    q1 = ops.maxpool2x2_fwd(y1, ops.Act(n, h // 2, w // 2, c))
    y2, q2, q3 = ops.Act(n, h, w, c), ops.Act(n, h // 2, w // 2, c), ops.Act(n, h // 2, w // 2, c)
    ops.bn_forward(za, _dev(gamma), _dev(beta), torch.zeros(c, device='cuda'), torch.ones(c, device='cuda'), st, y2, relu=True,
                   pooled=q2)
    ops.bn_forward(za, _dev(gamma), _dev(beta), torch.zeros(c, device='cuda'), torch.ones(c, device='cuda'), st, None, relu=True,
                   pooled=q3)
    torch.cuda.synchronize()
    assert torch.equal(y1.t, y2.t) and torch.equal(q1.t, q2.t) and torch.equal(q1.t, q3.t)


@pytest.mark.parametrize('shape', [(2, 16, 32, 64, 64), (3, 32, 64, 128, 256), (5, 96, 192, 64, 128)])
def test_conv_with_batch_statistics_in_its_epilogue(ops, shape):
    """xv_conv2d_fwd_stats (generation-4 conv kernel, STATS form): the conv output equals xv_conv2d_fwd's bit for bit and
    the per-channel sums it leaves equal xv_bn_stats on that output to fp32 summation order (integer operands: the sums of
    the stored values are exact in fp64, the partial sums in fp32 are not); shapes that do not tile are refused."""
    n, h, w, cin, cout = shape
    rng = np.random.default_rng(sum(shape))
    x = rng.integers(-2, 3, (n, h, w, cin)).astype(np.float32)
    wt = (rng.integers(-2, 3, (3, 3, cin, cout)) / 8.0).astype(np.float32)
    b = (rng.integers(-3, 4, cout) / 4.0).astype(np.float32)
    xa, wp, bd = ops.Act.from_dense(_dev(x)), ops.pack_conv_weights(_dev(wt)), _dev(b)
    z1, _ = ops.conv2d_fwd(xa, wp, bd, 3, relu=False)
    st1, st2 = ops.BnState(cout, 'cuda'), ops.BnState(cout, 'cuda')
    from modular_semantic_segmentation_amd import _lib
    _lib.check(_lib.lib().xv_bn_stats_ws(z1.xv(), st1.sums.data_ptr(), *st1.wsp(), None), 'xv_bn_stats_ws')
    z2 = ops.Act(n, h, w, cout)
    assert ops.conv2d_fwd_stats(xa, wp, bd, z2, st2)          # leaves per-workgroup rows; the sums are taken from them:
    lib = _lib.lib()
    _lib.check(lib.xv_bn_sums_from_rows(st2.conv_rows.data_ptr(), st2.conv_rows_n, 2 * cout, st2.sums.data_ptr(), None),
               'xv_bn_sums_from_rows')
    torch.cuda.synchronize()
    assert torch.equal(z1.t, z2.t)
    # ... or by the ONE launch that also finalises (xv_bn_finalize_from_rows): the same bits in sums, mean, invstd, scale,
    # shift and the moving statistics as the two calls; likewise xv_bn_stats_finalize_ws against xv_bn_stats_ws + finalize
    g = torch.Generator(device='cuda').manual_seed(3)
    gamma, beta = torch.rand(cout, device='cuda', generator=g) + 0.5, torch.randn(cout, device='cuda', generator=g)

    def fresh():
        st = ops.BnState(cout, 'cuda')
        return st, torch.zeros(cout, device='cuda') + 0.25, torch.ones(cout, device='cuda') * 2
    M = n * h * w
    sta, mma, mva = fresh()
    sta.sums.copy_(st2.sums)
    _lib.check(lib.xv_bn_finalize(sta.sums.data_ptr(), cout, M, gamma.data_ptr(), beta.data_ptr(), 1e-3, 0.99, mma.data_ptr(),
                                  mva.data_ptr(), sta.mean.data_ptr(), sta.invstd.data_ptr(), sta.scale.data_ptr(),
                                  sta.shift.data_ptr(), None), 'xv_bn_finalize')
    stb, mmb, mvb = fresh()
    _lib.check(lib.xv_bn_finalize_from_rows(st2.conv_rows.data_ptr(), st2.conv_rows_n, cout, M, gamma.data_ptr(), beta.data_ptr(),
                                            1e-3, 0.99, mmb.data_ptr(), mvb.data_ptr(), stb.mean.data_ptr(), stb.invstd.data_ptr(),
                                            stb.scale.data_ptr(), stb.shift.data_ptr(), stb.sums.data_ptr(), None),
               'xv_bn_finalize_from_rows')
    stc, mmc, mvc = fresh()
    _lib.check(lib.xv_bn_stats_finalize_ws(z1.xv(), stc.sums.data_ptr(), *stc.wsp(), gamma.data_ptr(), beta.data_ptr(), 1e-3,
                                           0.99, mmc.data_ptr(), mvc.data_ptr(), stc.mean.data_ptr(), stc.invstd.data_ptr(),
                                           stc.scale.data_ptr(), stc.shift.data_ptr(), None), 'xv_bn_stats_finalize_ws')
    std, mmd, mvd = fresh()
    std.sums.copy_(st1.sums)
    _lib.check(lib.xv_bn_finalize(std.sums.data_ptr(), cout, M, gamma.data_ptr(), beta.data_ptr(), 1e-3, 0.99, mmd.data_ptr(),
                                  mvd.data_ptr(), std.mean.data_ptr(), std.invstd.data_ptr(), std.scale.data_ptr(),
                                  std.shift.data_ptr(), None), 'xv_bn_finalize')
    torch.cuda.synchronize()
    for (a, ma, va), (b2, mb, vb) in (((sta, mma, mva), (stb, mmb, mvb)), ((std, mmd, mvd), (stc, mmc, mvc))):
        assert torch.equal(a.sums, b2.sums) and torch.equal(ma, mb) and torch.equal(va, vb)
        for name in ('mean', 'invstd', 'scale', 'shift'):
            assert torch.equal(getattr(a, name), getattr(b2, name)), name
    assert float(sta.scale.abs().sum()) > 0
    zs = z1.interior().double()
    want = torch.cat([zs.sum((0, 1, 2)), (zs * zs).sum((0, 1, 2))]).cpu().numpy()
    np.testing.assert_allclose(st2.sums.cpu().numpy(), want, rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(st1.sums.cpu().numpy(), want, rtol=1e-5, atol=1e-3)
    xs = ops.Act.from_dense(_dev(x[:, :h - 8]))
    assert not ops.conv2d_fwd_stats(xs, wp, bd, ops.Act(n, h - 8, w, cout), st2)          # 8 rows short of a tiling
