"""AdapNet expert: the gather kernels that put its strided / dilated convs on the MFMA
kernels, bit-exact on integers against plain torch convs with [TF1] padding, then the whole inference graph and the
model / fusion classes against oracle/adapnet_oracle.py."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import adapnet_oracle as ao
from oracle import fcn_oracle as fo

C, U = 12, 64
H, W = 64, 96


@pytest.fixture(scope='module')
def ops():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from modular_semantic_segmentation_amd import ops as _ops
    return _ops


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _nchw(x):
    return torch.from_numpy(x).permute(0, 3, 1, 2).contiguous()


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().numpy()


def _wt(w):
    return torch.from_numpy(w).permute(3, 2, 0, 1).contiguous()


def test_subsample2(ops):
    x = np.random.default_rng(0).integers(-8, 9, (2, 12, 20, 64)).astype(np.float32)
    y = ops.subsample2(ops.Act.from_dense(_dev(x)))
    torch.cuda.synchronize()
    assert np.array_equal(y.interior().float().cpu().numpy(), x[:, ::2, ::2])
    assert not y.t[:, 0].any() and not y.t[:, :, -1].any()


@pytest.mark.parametrize('n,h,w,pool', [(1, 16, 32, False), (2, 28, 44, True), (1, 64, 96, True)])
def test_conv7x7_stride2_through_gather_is_exact(ops, n, h, w, pool):
    """block_0_2 (adapnet.py:127): 7x7, stride 2, [TF1] 'same' = pad 2 before / 3 after."""
    from modular_semantic_segmentation_amd.adapnet import conv7s2_as_3x3
    rng = np.random.default_rng(h * w)
    x = rng.integers(-2, 3, (n, h, w, 64)).astype(np.float32)
    k7 = rng.integers(-1, 2, (7, 7, 64, 64)).astype(np.float32)
    b = rng.integers(-3, 4, 64).astype(np.float32)
    z = ops.gather_conv7s2(ops.Act.from_dense(_dev(x)))
    q = ops.Act(n, h // 4, w // 4, 64) if pool else None
    y, _ = ops.conv2d_fwd(z, ops.pack_conv_weights(_dev(conv7s2_as_3x3(k7))), _dev(b), 3, relu=True, pooled=q)
    torch.cuda.synchronize()
    ref = F.relu(F.conv2d(F.pad(_nchw(x), (2, 3, 2, 3)), _wt(k7), torch.from_numpy(b), stride=2))
    assert np.array_equal(y.interior().float().cpu().numpy(), fo.round_bf16(_nhwc(ref)))
    if pool:
        assert np.array_equal(q.interior().float().cpu().numpy(), fo.round_bf16(_nhwc(F.max_pool2d(ref, 2, 2))))


@pytest.mark.parametrize('n,h,w,cin', [(2, 32, 48, 3), (1, 16, 16, 1), (3, 6, 32, 3), (1, 2, 16, 3), (2, 64, 96, 1)])
def test_first_conv_written_as_the_7x7_operand(ops, n, h, w, cin):
    """xv_conv2d_first_gather7s2_fwd: block_0_1 (adapnet.py:126) stored directly as the operand of block_0_2 -- the same bytes
    as conv2d_first_fwd + gather_conv7s2, border and source-less positions included."""
    rng = np.random.default_rng(n * h + w + cin)
    x = _dev(rng.uniform(0, 255, (n, h, w, cin)).astype(np.float32))
    k = _dev((rng.standard_normal((3, 3, cin, 64)) * 0.05).astype(np.float32))
    b = _dev(rng.standard_normal(64).astype(np.float32))
    for relu in (True, False):
        y = ops.conv2d_first_fwd(x, k, b, ops.Act(n, h, w, 64), relu=relu)
        ref = ops.gather_conv7s2(y)
        z = ops.conv2d_first_gather7s2_fwd(x, k, b, ops.Act(n, h // 2, w // 2, 576), relu=relu)
        torch.cuda.synchronize()
        assert torch.equal(z.t, ref.t), (relu, float((z.t.float() - ref.t.float()).abs().max()))


@pytest.mark.parametrize('h,w,c,half,d1,d2', [(12, 20, 128, 32, 1, 2), (6, 10, 256, 128, 1, 16), (8, 12, 64, 32, 2, 4),
                                              (4, 6, 512, 256, 2, 16)])
def test_dilated_pair_through_im2col_is_exact(ops, h, w, c, half, d1, d2):
    """block_b's stage_2_1 / stage_2_2 + concat (adapnet.py:84-88), dilation rates up to beyond the image."""
    from modular_semantic_segmentation_amd.adapnet import dilated_pair_as_1x1
    rng = np.random.default_rng(h * w + d2)
    x = rng.integers(-2, 3, (2, h, w, c)).astype(np.float32)
    k1 = rng.integers(-1, 2, (3, 3, c, half)).astype(np.float32)
    k2 = rng.integers(-1, 2, (3, 3, c, half)).astype(np.float32)
    b = rng.integers(-3, 4, 2 * half).astype(np.float32)
    z = ops.im2col_dilated_pair(ops.Act.from_dense(_dev(x)), d1, d2)
    y, _ = ops.conv2d_fwd(z, ops.pack_conv_weights(_dev(dilated_pair_as_1x1(k1, k2))), _dev(b), 1, relu=True)
    torch.cuda.synchronize()
    xt = _nchw(x)
    ref = torch.cat([F.conv2d(xt, _wt(k1), torch.from_numpy(b[:half]), padding=d1, dilation=d1),
                     F.conv2d(xt, _wt(k2), torch.from_numpy(b[half:]), padding=d2, dilation=d2)], dim=1)
    assert np.array_equal(y.interior().float().cpu().numpy(), fo.round_bf16(_nhwc(F.relu(ref))))


@pytest.mark.parametrize('n,h,w,c,half,d1,d2', [(2, 12, 20, 64, 128, 1, 2), (2, 6, 10, 256, 128, 1, 16), (1, 24, 48, 128, 256, 2, 16),
                                                (3, 5, 7, 64, 128, 3, 5), (16, 24, 48, 64, 128, 2, 8), (2, 12, 20, 128, 32, 1, 2),
                                                (3, 5, 9, 64, 32, 2, 7)])
def test_dilated_pair_as_implicit_gemm(ops, n, h, w, c, half, d1, d2):
    """xv_conv_dilated_pair_fwd (no 18C operand, nine taps per output half gathered by the GEMM's loads): exact on integers
    against the two atrous convs + concat (adapnet.py:84-88), and bit for bit the materialised form on floats."""
    from modular_semantic_segmentation_amd.adapnet import dilated_pair_as_1x1
    rng = np.random.default_rng(h * w + d2)
    x = rng.integers(-2, 3, (n, h, w, c)).astype(np.float32)
    k1 = rng.integers(-1, 2, (3, 3, c, half)).astype(np.float32)
    k2 = rng.integers(-1, 2, (3, 3, c, half)).astype(np.float32)
    b = rng.integers(-3, 4, 2 * half).astype(np.float32)
    wp = ops.pack_conv_weights(_dev(dilated_pair_as_1x1(k1, k2)))
    assert ops.dilated_pair_implicit_ok(c, 2 * half)
    y = ops.conv_dilated_pair(ops.Act.from_dense(_dev(x)), wp, _dev(b), d1, d2, relu=True)
    torch.cuda.synchronize()
    xt = _nchw(x)
    ref = torch.cat([F.conv2d(xt, _wt(k1), torch.from_numpy(b[:half]), padding=d1, dilation=d1),
                     F.conv2d(xt, _wt(k2), torch.from_numpy(b[half:]), padding=d2, dilation=d2)], dim=1)
    assert np.array_equal(y.interior().float().cpu().numpy(), fo.round_bf16(_nhwc(F.relu(ref))))
    assert float(y.t.float().abs().sum()) == float(y.interior().float().abs().sum())      # the border stays zero
    # floats: the same products in the same order as im2col + 1x1 conv (whose other nine taps add exact zeros)
    xf = ops.Act.from_dense(_dev(rng.standard_normal((n, h, w, c)).astype(np.float32)))
    wf = ops.pack_conv_weights(_dev(dilated_pair_as_1x1(rng.standard_normal(k1.shape).astype(np.float32) * 0.05,
                                                        rng.standard_normal(k2.shape).astype(np.float32) * 0.05)))
    bf = _dev(rng.standard_normal(2 * half).astype(np.float32))
    for relu in (True, False):
        a = ops.conv_dilated_pair(xf, wf, bf, d1, d2, relu=relu)
        m, _ = ops.conv2d_fwd(ops.im2col_dilated_pair(xf, d1, d2), wf, bf, 1, relu=relu)
        assert torch.equal(a.t, m.t), relu


@pytest.mark.parametrize('n,h,w,c,half,d1,d2', [(2, 12, 20, 256, 128, 1, 2), (1, 24, 48, 256, 128, 2, 16), (2, 6, 10, 512, 256, 1, 16),
                                                (3, 5, 7, 256, 128, 3, 5), (8, 24, 48, 256, 128, 1, 4)])
def test_dilated_pair_gradients_without_the_operand(ops, n, h, w, c, half, d1, d2):
    """xv_conv_dilated_pair_bwd_data / _bwd_filter_ws: the gradients of the two atrous convs + concat (adapnet.py:84-88) with the
    taps gathered by the kernels' loads -- exact on integers against autograd, accumulated INTO dw, bitwise reproducible."""
    rng = np.random.default_rng(h * w + d2 + c)
    x = rng.integers(-2, 3, (n, h, w, c)).astype(np.float32)
    k1 = rng.integers(-1, 2, (3, 3, c, half)).astype(np.float32)
    k2 = rng.integers(-1, 2, (3, 3, c, half)).astype(np.float32)
    dy = (rng.integers(-1, 2, (n, h, w, 2 * half)) * (rng.random((n, h, w, 2 * half)) < 0.1)).astype(np.float32)
    xt = _nchw(x).requires_grad_(True)
    w1, w2 = _wt(k1).requires_grad_(True), _wt(k2).requires_grad_(True)
    y = torch.cat([F.conv2d(xt, w1, padding=d1, dilation=d1), F.conv2d(xt, w2, padding=d2, dilation=d2)], dim=1)
    y.backward(_nchw(dy))
    xa, dya = ops.Act.from_dense(_dev(x)), ops.Act.from_dense(_dev(dy))
    assert ops.dilated_pair_training_ok(xa, 2 * half)
    # data gradient
    wd = ops.pack_conv_weights(ops.dilated_pair_dgrad_kernel(_dev(k1), _dev(k2)))
    dx = ops.conv_dilated_pair_bwd_data(dya, wd, torch.zeros(c, device='cuda'), d1, d2, ops.Act(n, h, w, c))
    torch.cuda.synchronize()
    assert np.array_equal(dx.interior().float().cpu().numpy(), fo.round_bf16(_nhwc(xt.grad)))
    assert float(dx.t.float().abs().sum()) == float(dx.interior().float().abs().sum())
    # filter gradients
    ws = torch.empty(ops.conv_dilated_pair_bwd_filter_workspace_bytes(xa, 2 * half) // 4, device='cuda').fill_(float('nan'))
    dw1, dw2 = torch.ones(3, 3, c, half, device='cuda'), torch.full((3, 3, c, half), -2.0, device='cuda')
    ops.conv_dilated_pair_bwd_filter(xa, dya, d1, d2, dw1, dw2, ws)
    torch.cuda.synchronize()
    assert np.array_equal(dw1.cpu().numpy(), w1.grad.permute(2, 3, 1, 0).numpy() + 1)
    assert np.array_equal(dw2.cpu().numpy(), w2.grad.permute(2, 3, 1, 0).numpy() - 2)
    # floats: the same bits from run to run, and the materialised form's sums to rounding
    xf = ops.Act.from_dense(_dev(rng.standard_normal((n, h, w, c)).astype(np.float32)))
    df = ops.Act.from_dense(_dev(rng.standard_normal((n, h, w, 2 * half)).astype(np.float32)))
    outs = []
    for _ in range(2):
        a, b = torch.zeros(3, 3, c, half, device='cuda'), torch.zeros(3, 3, c, half, device='cuda')
        ops.conv_dilated_pair_bwd_filter(xf, df, d1, d2, a, b, ws)
        outs.append((a, b))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    z = ops.im2col_dilated_pair(xf, d1, d2)
    full = torch.zeros(1, 1, 18 * c, 2 * half, device='cuda')
    ws1 = torch.empty(ops.conv2d_bwd_filter_workspace_bytes(z, 2 * half, 1) // 4, device='cuda')
    ops.conv2d_bwd_filter(z, df, full, None, 1, workspace=ws1)
    want1, want2 = full[0, 0, :9 * c, :half].reshape(3, 3, c, half), full[0, 0, 9 * c:, half:].reshape(3, 3, c, half)
    scale = float(full.abs().max())
    assert float((outs[0][0] - want1).abs().max()) <= 2e-5 * scale and float((outs[0][1] - want2).abs().max()) <= 2e-5 * scale


def test_dilated_pair_implicit_rejects_other_shapes(ops):
    from modular_semantic_segmentation_amd import _lib
    x = ops.Act(1, 4, 4, 64)
    assert not ops.dilated_pair_implicit_ok(64, 128) and not ops.dilated_pair_implicit_ok(32, 256) and ops.dilated_pair_implicit_ok(128, 64)
    with pytest.raises(_lib.XvError):
        ops.conv_dilated_pair(x, torch.zeros(18 * 64 * 128, dtype=torch.bfloat16, device='cuda'),
                              torch.zeros(128, device='cuda'), 1, 2)


def test_residual_conv_and_affine_upsample(ops):
    rng = np.random.default_rng(3)
    x = rng.integers(-2, 3, (2, 10, 14, 128)).astype(np.float32)
    k = rng.integers(-1, 2, (1, 1, 128, 256)).astype(np.float32)
    b = rng.integers(-3, 4, 256).astype(np.float32)
    r = rng.integers(0, 9, (2, 10, 14, 256)).astype(np.float32)
    y = ops.conv1x1_residual(ops.Act.from_dense(_dev(x)), ops.pack_conv_weights(_dev(k)), _dev(b),
                             ops.Act.from_dense(_dev(r)), relu=True)
    torch.cuda.synchronize()
    ref = F.relu(F.conv2d(_nchw(x), _wt(k), torch.from_numpy(b))) + _nchw(r)
    assert np.array_equal(y.interior().float().cpu().numpy(), fo.round_bf16(_nhwc(ref)))
    # x2 bilinear + batch norm without activation + residual (adapnet.py:157-163)
    xs = rng.integers(-4, 5, (1, 6, 8, 64)).astype(np.float32)
    res = rng.integers(-4, 5, (1, 12, 16, 64)).astype(np.float32)
    s = rng.uniform(-1.5, 1.5, 64).astype(np.float32)
    t = rng.uniform(-1, 1, 64).astype(np.float32)
    got = ops.upsample2x_relu_add(ops.Act.from_dense(_dev(xs)), residual=ops.Act.from_dense(_dev(res)),
                                  scale=_dev(s), shift=_dev(t), relu=False)
    torch.cuda.synchronize()
    up = fo.deconv_same(_nchw(xs), fo.bilinear_kernel(4, 64), 2)
    want = _nhwc(up) * s + t + res
    assert np.abs(got.interior().float().cpu().numpy() - want).max() <= 2 ** -7 * np.abs(want).max()
    assert (want < 0).any()


def test_gather_transposes_are_exact_adjoints(ops):
    """<G x, dz> == <x, G^T dz> on small integers (exact in bf16 / fp32) for the three gathers, borders untouched;
    the residual add."""
    rng = np.random.default_rng(11)

    def ints(shape, lo=-2, hi=3):
        return rng.integers(lo, hi, shape).astype(np.float32)

    def dot(a, b):
        return float((a.interior().double() * b.interior().double()).sum().item())

    def clean_border(a):
        return not (a.t[:, 0].any() or a.t[:, -1].any() or a.t[:, :, 0].any() or a.t[:, :, -1].any())

    x = ops.Act.from_dense(_dev(ints((2, 12, 20, 64))))
    dy = ops.Act.from_dense(_dev(ints((2, 6, 10, 64))))
    dx = ops.subsample2_bwd(dy, ops.Act(2, 12, 20, 64))
    assert dot(ops.subsample2(x), dy) == dot(x, dx) and clean_border(dx)
    assert np.array_equal(dx.interior().float().cpu().numpy()[:, ::2, ::2], dy.interior().float().cpu().numpy())

    for h, w in ((12, 20), (2, 4), (16, 6)):
        x = ops.Act.from_dense(_dev(ints((2, h, w, 64))))
        dz = ops.Act.from_dense(_dev(ints((2, h // 2, w // 2, 576))))
        dx = ops.gather_conv7s2_bwd(dz, ops.Act(2, h, w, 64))
        assert dot(ops.gather_conv7s2(x), dz) == dot(x, dx) and clean_border(dx), (h, w)

    for h, w, c, d1, d2 in ((12, 20, 64, 1, 2), (6, 10, 128, 2, 16), (3, 5, 64, 4, 8)):
        x = ops.Act.from_dense(_dev(ints((2, h, w, c))))
        dz = ops.Act.from_dense(_dev(ints((2, h, w, 18 * c))))
        dx = ops.im2col_dilated_pair_bwd(dz, d1, d2, ops.Act(2, h, w, c))
        assert dot(ops.im2col_dilated_pair(x, d1, d2), dz) == dot(x, dx) and clean_border(dx), (h, w, d1, d2)

    a, b = ints((1, 5, 7, 64), -30, 30), ints((1, 5, 7, 64), -30, 30)
    y = ops.add(ops.Act.from_dense(_dev(a)), ops.Act.from_dense(_dev(b)))
    torch.cuda.synchronize()
    assert np.array_equal(y.interior().float().cpu().numpy(), a + b) and clean_border(y)


def _weights(tmp_path, prefix, cin, seed, scale_first):
    w = ao.init_adapnet_weights(prefix, cin, U, C, seed=seed, gain=1.3)
    w['%s/block_0_1/kernel' % prefix] *= scale_first
    path = os.path.join(str(tmp_path), prefix + '_adapnet.npz')
    np.savez(path, **w)
    return w, path


def _inputs(n, seed=0):
    rng = np.random.default_rng(seed)
    return {'rgb': rng.integers(0, 256, (n, H, W, 3)).astype(np.float32),
            'depth': rng.integers(0, 65536, (n, H, W, 1)).astype(np.float32),
            'labels': rng.integers(-1, C, (n, H, W)).astype(np.int32)}


def _desc():
    return ({'rgb': 'float32', 'depth': 'float32', 'labels': 'int32'},
            {'rgb': (None, None, 3), 'depth': (None, None, 1), 'labels': (None, None)}, C)


def test_adapnet_graph_matches_oracle(ops, tmp_path):
    from modular_semantic_segmentation_amd.adapnet import AdapnetEngine, variable_shapes
    w, _ = _weights(tmp_path, 'rgb', 3, 1, 0.02)
    assert {k: tuple(v.shape) for k, v in w.items()} == variable_shapes('rgb', 3, U, C)
    data = _inputs(2)
    eng = AdapnetEngine('rgb', 3, U, C, w)
    out = eng.forward(_dev(data['rgb']), want=('score', 'prob', 'label'))
    torch.cuda.synchronize()
    ref = ao.adapnet_forward(data['rgb'], w, 'rgb', policy='bf16',
                             keep=['block_0_pool', 'block_3', 'block_7', 'block_13', 'block_16', 'shortcut', 'merge', 'score'])
    for name in ('block_0_pool', 'block_3', 'block_7', 'block_13', 'block_16', 'shortcut', 'merge'):
        got = out['layers'][name].interior().float().cpu().numpy()[..., :ref[name].shape[-1]]
        err = np.abs(got - ref[name]).max() / np.abs(ref[name]).max()
        # bf16 storage of 50 chained layers: a few ulps (2^-8) of the largest activation
        assert err < 3e-2, '%s differs by %.3g of its max' % (name, err)
    score = out['score'].cpu().numpy()
    scale = np.abs(ref['score']).max()
    assert np.abs(score - ref['score']).max() / scale < 3e-2
    label = out['label'].cpu().numpy()
    # labels: bit-exact against the oracle's softmax + argmax of the SAME logits, and agreeing with the oracle's own
    # wherever its decision is clear
    assert np.array_equal(label, fo.argmax_last(fo.softmax(score)))
    ref_label = fo.argmax_last(fo.softmax(ref['score']))
    top2 = np.sort(ref['score'], -1)[..., -2:]
    clear = (top2[..., 1] - top2[..., 0]) > 6e-2 * scale
    assert clear.mean() > 0.3 and np.array_equal(label[clear], ref_label[clear])
    assert (label == ref_label).mean() > 0.95
    assert np.allclose(out['prob'].cpu().numpy(), fo.softmax(score), atol=1e-5)
    assert len(np.unique(ref_label)) > 2


@pytest.mark.parametrize('which', ['both', 'first', 'second'])
def test_adapnet_with_trained_deconv_kernels_matches_oracle(ops, tmp_path, which):
    """A checkpoint the REFERENCE trained holds DENSE deconv kernels: adapnet.py:155-163 calls custom_layers.deconv2d
    (:71-121) without trainable=False, so `first_/second_deconvolution_upconv/kernel` leave the bilinear constant at the
    first optimizer step.  Such an npz is imported through the model API and evaluated on the dense transposed-conv path
    (3x3 MFMA conv onto the stride^2 output phases + depth-to-space), against the oracle's conv_transpose2d; either
    kernel alone may be dense (the other keeps its depthwise fast path)."""
    from modular_semantic_segmentation_amd import get_model
    from modular_semantic_segmentation_amd.adapnet import rect_bilinear_filter
    w, _ = _weights(tmp_path, 'rgb', 3, 1, 0.02)
    rng = np.random.default_rng(11)
    for scope, on in (('first_deconvolution_upconv', which in ('both', 'first')),
                      ('second_deconvolution_upconv', which in ('both', 'second'))):
        if on:
            k = w['rgb/%s/kernel' % scope]
            # "trained-looking": the bilinear constant plus a dense perturbation of a comparable size per output
            fan = k.shape[3]
            w['rgb/%s/kernel' % scope] = (k + rng.standard_normal(k.shape) * (0.3 / np.sqrt(fan))).astype(np.float32)
            assert not np.allclose(w['rgb/%s/kernel' % scope], rect_bilinear_filter(k.shape))
        for v, lo, hi in (('gamma', 0.7, 1.3), ('beta', -0.2, 0.2), ('moving_mean', -0.2, 0.2), ('moving_variance', 0.5, 1.5)):
            shape = w['rgb/%s/%s' % (scope, v)].shape
            w['rgb/%s/%s' % (scope, v)] = rng.uniform(lo, hi, shape).astype(np.float32)
    path = os.path.join(str(tmp_path), 'trained_adapnet.npz')
    np.savez(path, **w)
    data = _inputs(2, seed=3)
    with get_model('adapnet')(_desc(), output_dir=str(tmp_path), modality='rgb', num_units=U, batchsize=2) as net:
        net.import_weights(path)
        assert set(net.engine.dense) == {'both': {'first', 'second'}, 'first': {'first'}, 'second': {'second'}}[which]
        score = net.predict(data, output_attr='score')
        label = net.predict(data)
        prob = net.predict(data, output_attr='prob')
        merge = net.engine.trunk(_dev(data['rgb']))['merge'].interior().float().cpu().numpy()
    ref = ao.adapnet_forward(data['rgb'], w, 'rgb', policy='bf16', keep=['merge', 'score'])
    err = np.abs(merge[..., :U] - ref['merge']).max() / np.abs(ref['merge']).max()
    assert err < 3e-2, 'merge differs by %.3g of its max' % err
    scale = np.abs(ref['score']).max()
    print('dense deconv (%s): max logit error %.4f of scale' % (which, np.abs(score - ref['score']).max() / scale))
    assert np.abs(score - ref['score']).max() / scale < 3e-2
    assert np.array_equal(label, fo.argmax_last(fo.softmax(score)))
    assert np.allclose(prob, fo.softmax(score), atol=1e-5)
    ref_label = fo.argmax_last(fo.softmax(ref['score']))
    top2 = np.sort(ref['score'], -1)[..., -2:]
    clear = (top2[..., 1] - top2[..., 0]) > 6e-2 * scale
    assert np.array_equal(label[clear], ref_label[clear])
    # fp32 graph too: the dense kernels are honoured, not replaced by the bilinear constant (which would be far off)
    ref32 = ao.adapnet_forward(data['rgb'], w, 'rgb', policy='fp32', keep=['score'])['score']
    wb = dict(w)
    for scope in ('first_deconvolution_upconv', 'second_deconvolution_upconv'):
        wb['rgb/%s/kernel' % scope] = rect_bilinear_filter(w['rgb/%s/kernel' % scope].shape)
    refb = ao.adapnet_forward(data['rgb'], wb, 'rgb', policy='fp32', keep=['score'])['score']
    assert np.abs(score - ref32).max() < 0.25 * np.abs(refb - ref32).max()


@pytest.mark.parametrize('n,h,w,cin', [(1, 16, 32, 1), (3, 48, 80, 3)])
def test_adapnet_small_and_odd_shapes(ops, tmp_path, n, h, w, cin):
    """Maps down to 1x2 pixels at stride 16 (every atrous tap but the centre off the image), a depth-style one-channel
    input, batch sizes that do not fill a GEMM row tile."""
    from modular_semantic_segmentation_amd.adapnet import AdapnetEngine
    w_, _ = _weights(tmp_path, 'm', cin, 7, 0.02)
    x = np.random.default_rng(h).integers(0, 256, (n, h, w, cin)).astype(np.float32)
    out = AdapnetEngine('m', cin, U, C, w_).forward(_dev(x), want=('score', 'label'))
    torch.cuda.synchronize()
    ref = ao.adapnet_forward(x, w_, 'm', policy='bf16')
    score = out['score'].cpu().numpy()
    assert np.abs(score - ref['score']).max() / np.abs(ref['score']).max() < 3e-2
    assert np.array_equal(out['label'].cpu().numpy(), fo.argmax_last(fo.softmax(score)))


def test_adapnet_model_and_fusion_classes(ops, tmp_path):
    from modular_semantic_segmentation_amd import get_model
    w_rgb, p_rgb = _weights(tmp_path, 'rgb', 3, 1, 0.02)
    w_dep, p_dep = _weights(tmp_path, 'depth', 1, 2, 2e-4)
    data = _inputs(3, seed=5)
    with get_model('adapnet')(_desc(), output_dir=str(tmp_path), modality='rgb', num_units=U, batchsize=2) as net:
        net.import_weights(p_rgb)
        pred = net.predict(data)
        assert pred.shape == (3, H, W) and pred.dtype == np.int64
        ref = ao.adapnet_forward(data['rgb'], w_rgb, 'rgb', policy='bf16')['score']
        assert (pred == fo.argmax_last(fo.softmax(ref))).mean() > 0.95
        measures, cm = net.score(data)
        assert cm.sum() == (data['labels'] >= 0).sum()
        exported = np.load(net.export_weights())
        assert np.array_equal(exported['rgb/block_layer_7/stage_2_2/kernel'], w_rgb['rgb/block_layer_7/stage_2_2/kernel'])
    cms = {m: np.eye(C) * 50 + 1 for m in ('rgb', 'depth')}
    with get_model('bayes_fusion')(data_description=_desc(), confusion_matrices=cms,
                                   prefixes={'rgb': 'rgb', 'depth': 'depth'}, num_units=U,
                                   num_channels={'rgb': 3, 'depth': 1}, expert_model='adapnet', class_prior='uniform',
                                   batchsize=3) as net:
        net.import_weights(p_rgb)
        net.import_weights(p_dep)
        fused = net.predict(data)
        labels = {m: fo.argmax_last(fo.softmax(ao.adapnet_forward(data[m], w, m, policy='bf16')['score']))
                  for m, w in (('rgb', w_rgb), ('depth', w_dep))}
        # identical diagonal-dominant confusion matrices + uniform prior: where the experts agree, so does the fusion
        agree = labels['rgb'] == labels['depth']
        assert agree.any() and (fused[agree] == labels['rgb'][agree]).mean() > 0.95


def test_dirichlet_fusion_with_adapnet_experts(ops, tmp_path):
    """DirichletFusion's expert choice goes through the same test_pipeline switch (dirichlet_mix.py:96-99)."""
    from modular_semantic_segmentation_amd import get_model
    _, p_rgb = _weights(tmp_path, 'rgb', 3, 1, 0.02)
    _, p_dep = _weights(tmp_path, 'depth', 1, 2, 2e-4)
    data = _inputs(4, seed=9)
    cfg = {'expert_model': 'adapnet', 'modalities': ['rgb', 'depth'], 'num_units': U,
           'num_channels': {'rgb': 3, 'depth': 1}, 'class_prior': 'data', 'sigma': 1.0, 'delta': 1e-2, 'beta': 1e-2,
           'batchsize': 2}
    with get_model('dirichlet_fusion')(data_description=_desc(), **cfg) as net:
        net.import_weights(p_rgb)
        net.import_weights(p_dep)
        params = net.fit(data)
        assert params['rgb'].shape == (C, C) and np.all(params['rgb'] > 0)
        net.import_weights(p_rgb)
        net.import_weights(p_dep)
        measures, cm = net.score(data)
        assert cm.sum() == (data['labels'] >= 0).sum()
    with pytest.raises(UserWarning):
        get_model('bayes_fusion')(data_description=_desc(), confusion_matrices={m: np.eye(C) for m in ('rgb', 'depth')},
                                  prefixes={'rgb': 'rgb', 'depth': 'depth'}, num_units=U,
                                  num_channels={'rgb': 3, 'depth': 1}, expert_model='resnet', class_prior='uniform')


SHALLOW = [('block_layer_1', 'a', (64, 128, 1, True)), ('block_layer_2', 'a', (64, 128, 1, False)),
           ('block_layer_4', 'a', (64, 256, 2, True)), ('block_layer_7', 'b', (64, 64, 256, 1, 2, False)),
           ('block_layer_8', 'a', (128, 512, 2, True)), ('block_layer_14', 'b', (128, 128, 512, 2, 4, False))]


def _grad_agreement(got, ref_g):
    rel, cos = {}, {}
    for k, g in ref_g.items():
        if k.endswith('/bias'):
            continue                                   # a bias in front of a batch norm has an exactly zero gradient
        a = got[k][..., :g.shape[-1]].ravel().astype(np.float64)
        b = g.ravel().astype(np.float64)
        rel[k] = np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30)
        cos[k] = a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30)
    return rel, cos


WIDE = [('block_layer_1', 'a', (64, 128, 1, True)), ('block_layer_4', 'a', (64, 256, 2, True)),
        ('block_layer_7', 'b', (64, 64, 256, 1, 2, False)), ('block_layer_8', 'a', (128, 512, 2, True)),
        ('block_layer_10', 'b', (256, 256, 512, 2, 4, False))]


def test_adapnet_training_step_with_the_implicit_atrous_pair(ops):
    """block_b with 256-channel pairs trains WITHOUT the im2col operand (ops.conv_dilated_pair + _bwd_filter + _bwd_data);
    the same step through the operand (adapnet_trainer._IMPLICIT_PAIRS = False) must give the same loss, the same pair-kernel
    gradients up to the order of the fp32 sums, and the other gradients up to the bf16 rounding of the pair's data gradient
    (the operand form rounds every tap's contribution to bf16 before adding them, the implicit one rounds the sum)."""
    from modular_semantic_segmentation_amd import adapnet_trainer
    from modular_semantic_segmentation_amd.adapnet import AdapnetEngine
    h, w = 64, 96
    rng = np.random.default_rng(5)
    x = rng.integers(0, 256, (2, h, w, 3)).astype(np.float32)
    labels = rng.integers(-1, C, (2, h, w)).astype(np.int32)
    w_ = ao.init_adapnet_weights('rgb', 3, U, C, seed=2, gain=1.3, blocks=WIDE)
    w_['rgb/block_0_1/kernel'] *= 0.02
    results = {}
    for implicit in (True, False):
        adapnet_trainer._IMPLICIT_PAIRS = implicit
        try:
            eng = AdapnetEngine('rgb', 3, U, C, w_, blocks=WIDE)
            tr = adapnet_trainer.AdapnetTrainer(eng, 'rmsprop', 1e-3)
            tr.load_from_variables(w_)
            assert ('block_layer_10' in tr.wd_pair) == implicit and 'block_layer_7' not in tr.wd_pair
            loss = tr.step(_dev(x), _dev(labels))
            torch.cuda.synchronize()
            results[implicit] = (loss.item(), tr.grads_as_variables())
        finally:
            adapnet_trainer._IMPLICIT_PAIRS = True
    (la, ga), (lb, gb) = results[True], results[False]
    assert la == lb                                        # the forward passes are the same bits
    rel, cos = _grad_agreement(ga, gb)
    for k in ('rgb/block_layer_10/stage_2_1/kernel', 'rgb/block_layer_10/stage_2_2/kernel'):
        assert rel[k] < 1e-4, (k, rel[k])                  # same dz, same map: only the order of the fp32 sums differs
    worst = max(rel.items(), key=lambda kv: kv[1])
    assert worst[1] < 0.1 and min(cos.values()) > 0.995, (worst, min(cos.items(), key=lambda kv: kv[1]))


def test_adapnet_training_step_on_a_shallow_graph(ops):
    """Every op type of the training graph -- first conv, 7x7 stride-2 through the gather, pool, block_a with and
    without shortcut conv and with stride 2, block_b with its stacked atrous pair, the block-7 shortcut, both deconvs
    with batch norm, the head -- on a 6-block graph, shallow enough that bf16 rounding noise does not drown the
    comparison: perturbing the oracle's own kernels by 3e-7 moves ITS gradients by 25-40 % here (cosine 0.93+), against
    100 % (cosine 0.25-0.5) on the full 16 blocks.  A wrong term, index map or mask shows up as a cosine far below
    that."""
    from modular_semantic_segmentation_amd.adapnet import AdapnetEngine
    from modular_semantic_segmentation_amd.adapnet_trainer import AdapnetTrainer
    h, w = 64, 96
    rng = np.random.default_rng(0)
    x = rng.integers(0, 256, (2, h, w, 3)).astype(np.float32)
    labels = rng.integers(-1, C, (2, h, w)).astype(np.int32)
    w_ = ao.init_adapnet_weights('rgb', 3, U, C, seed=1, gain=1.3, blocks=SHALLOW)
    w_['rgb/block_0_1/kernel'] *= 0.02
    eng = AdapnetEngine('rgb', 3, U, C, w_, blocks=SHALLOW)
    # the inference graph of the same shallow net first
    ref = ao.adapnet_forward(x, w_, 'rgb', policy='bf16', blocks=SHALLOW)['score']
    score = eng.forward(_dev(x), want=('score',))['score'].cpu().numpy()
    assert np.abs(score - ref).max() / np.abs(ref).max() < 2e-2
    tr = AdapnetTrainer(eng, 'rmsprop', 1e-3)
    tr.load_from_variables(w_)
    loss = tr.step(_dev(x), _dev(labels))
    torch.cuda.synchronize()
    got = tr.grads_as_variables()
    ref_loss, ref_g, stats = ao.adapnet_loss_and_grads(x, labels, w_, 'rgb', C, policy='bf16', blocks=SHALLOW)
    assert abs(loss.item() - ref_loss) < 1e-2 * abs(ref_loss), (loss.item(), ref_loss)
    assert set(got) == set(ref_g)
    rel, cos = _grad_agreement(got, ref_g)
    print('shallow adapnet: gradient vs the bf16-policy oracle: relative error, cosine')
    for k in sorted(rel):
        print('  %-50s %.4f %.4f' % (k, rel[k], cos[k]))
    for k in ('rgb/second_deconvolution_upconv/gamma', 'rgb/second_deconvolution_upconv/beta'):
        assert rel[k] < 0.05, (k, rel[k])
    for k in ('rgb/shortcut/kernel', 'rgb/first_deconvolution_conv/kernel', 'rgb/first_deconvolution_upconv/gamma'):
        assert rel[k] < 0.2, (k, rel[k])
    # the two TRAINED transposed-conv kernels (adapnet.py:155-163): the x8 one sits right under the loss
    assert rel['rgb/second_deconvolution_upconv/kernel'] < 0.05, rel['rgb/second_deconvolution_upconv/kernel']
    assert rel['rgb/first_deconvolution_upconv/kernel'] < 0.2, rel['rgb/first_deconvolution_upconv/kernel']
    assert min(cos.values()) > 0.85, min(cos.items(), key=lambda kv: kv[1])
    assert max(rel.values()) < 0.7, max(rel.items(), key=lambda kv: kv[1])
    out = dict(w_)
    tr.to_variables(out)
    for scope in ('block_0_1', 'block_0_2', 'block_layer_4/stage_1', 'block_layer_7/stage_2_2', 'block_layer_14/stage_2_1',
                  'shortcut', 'first_deconvolution_upconv', 'second_deconvolution_upconv'):
        mean, var = stats[scope]
        real = len(w_['rgb/%s/moving_mean' % scope])
        np.testing.assert_allclose(out['rgb/%s/moving_mean' % scope],
                                   0.99 * w_['rgb/%s/moving_mean' % scope] + 0.01 * mean[:real], rtol=3e-2, atol=3e-3)
        np.testing.assert_allclose(out['rgb/%s/moving_variance' % scope],
                                   0.99 * w_['rgb/%s/moving_variance' % scope] + 0.01 * var[:real], rtol=3e-2, atol=3e-3)
    # RMSProp ([TF1]: ms starts at 1) with the loss's 1/count already in the gradient
    k = 'rgb/block_layer_7/stage_2_1/kernel'
    g = got[k]
    np.testing.assert_allclose(out[k] - w_[k], -1e-3 * g / np.sqrt(0.9 + 0.1 * g * g + 1e-10), rtol=2e-3, atol=4e-9)


def test_adapnet_score_deconv_is_float32_in_inference_and_training(ops):
    """The trained x8 deconv of the class scores (adapnet.py:155-163, float32 in the reference): on the SAME `merge` map the
    device scores equal a float32 conv_transpose2d with the trained kernel to 1e-5 of the score scale -- fp32 summation order
    only -- in the inference engine and in the training step (round 4 sent them through a bf16 phase map: 4e-3)."""
    from modular_semantic_segmentation_amd.adapnet import AdapnetEngine
    from modular_semantic_segmentation_amd.adapnet_trainer import AdapnetTrainer
    from modular_semantic_segmentation_amd.fcn import BN_EPS
    h, w = 64, 96
    rng = np.random.default_rng(3)
    x = rng.integers(0, 256, (2, h, w, 3)).astype(np.float32)
    labels = rng.integers(-1, C, (2, h, w)).astype(np.int32)
    w_ = ao.init_adapnet_weights('rgb', 3, U, C, seed=2, gain=1.3, blocks=SHALLOW)
    w_['rgb/block_0_1/kernel'] *= 0.02
    k = w_['rgb/second_deconvolution_upconv/kernel']
    w_['rgb/second_deconvolution_upconv/kernel'] = (k + rng.standard_normal(k.shape) * (0.3 / np.sqrt(k.shape[3]))).astype(np.float32)
    kern = w_['rgb/second_deconvolution_upconv/kernel']

    def reference(merge_act):
        m = merge_act.interior().float().cpu()[..., :U].permute(0, 3, 1, 2).contiguous()
        return fo.deconv_same(m, kern, 8).permute(0, 2, 3, 1).numpy()

    eng = AdapnetEngine('rgb', 3, U, C, w_, blocks=SHALLOW)
    assert 'second' in eng.dense
    out = eng.forward(_dev(x), want=('score',))
    torch.cuda.synchronize()
    g = w_['rgb/second_deconvolution_upconv/gamma'] / np.sqrt(w_['rgb/second_deconvolution_upconv/moving_variance'] + BN_EPS)
    t = w_['rgb/second_deconvolution_upconv/beta'] - w_['rgb/second_deconvolution_upconv/moving_mean'] * g
    ref = reference(out['layers']['merge']) * g + t
    err = np.abs(out['score'].cpu().numpy() - ref).max() / np.abs(ref).max()
    print('inference: x8 score deconv vs float32 on the same merge map: %.2e of the scale' % err)
    assert err < 1e-5
    tr = AdapnetTrainer(eng, 'rmsprop', 1e-3)
    tr.load_from_variables(w_)
    tr.step(_dev(x), _dev(labels))
    torch.cuda.synchronize()
    merge = [a for key, a in tr._a.items() if key[0] == 'merge'][0]
    raw = tr._dense('score_raw', (2, h, w, C)).cpu().numpy()
    ref = reference(merge)
    err = np.abs(raw - ref).max() / np.abs(ref).max()
    print('training: raw scores vs float32 on the same merge map: %.2e of the scale' % err)
    assert err < 1e-5


def test_adapnet_training_step(ops, tmp_path):
    """One training step (adapnet.py:103-173 with is_training=True, loss :196-203) against autograd over the oracle's
    restatement: loss, gradients of every kernel / bias / gamma / beta, moving averages; then fit(), export and
    inference with the trained statistics.  A 50-layer batch-norm network at random initialisation is chaotic with
    respect to bf16 rounding (see tests/test_backward_gpu.py); thresholds are calibrated against the oracle's own
    sensitivity, the kernels on the path are each checked exactly elsewhere in this file."""
    from modular_semantic_segmentation_amd import get_model
    h, w = 64, 96
    rng = np.random.default_rng(0)
    data = {'rgb': rng.integers(0, 256, (2, h, w, 3)).astype(np.float32),
            'labels': rng.integers(-1, C, (2, h, w)).astype(np.int32)}
    w_, path = _weights(tmp_path, 'rgb', 3, 1, 0.02)
    net = get_model('adapnet')(_desc(), output_dir=str(tmp_path), modality='rgb', num_units=U, batchsize=2,
                               trainer='adam', learning_rate=1e-3)
    net.import_weights(path)
    tr = net._ensure_trainer()
    loss = tr.step(_dev(data['rgb']), _dev(data['labels']))
    torch.cuda.synchronize()
    got = tr.grads_as_variables()
    ref_loss, ref_g, stats = ao.adapnet_loss_and_grads(data['rgb'], data['labels'], w_, 'rgb', C, policy='bf16')
    assert abs(loss.item() - ref_loss) < 2e-2 * abs(ref_loss), (loss.item(), ref_loss)
    assert set(got) == set(ref_g)
    rel, cos = _grad_agreement(got, ref_g)
    # full depth: the oracle's own gradients move by ~100 % (cosine 0.25-0.5) under a 3e-7 perturbation of its kernels,
    # so only the head is compared here; the backward graph itself is pinned on the shallow graph above
    print('median relative error %.4f, worst cosine %.4f' % (np.median(list(rel.values())), min(cos.values())))
    assert rel['rgb/second_deconvolution_upconv/gamma'] < 0.1 and rel['rgb/second_deconvolution_upconv/beta'] < 0.3
    assert np.median(list(rel.values())) < 1.3
    out = dict(net.variables)
    tr.to_variables(out)
    for scope in ('block_0_1', 'block_0_2', 'block_layer_1/stage_2', 'second_deconvolution_upconv'):
        mean, var = stats[scope]
        np.testing.assert_allclose(out['rgb/%s/moving_mean' % scope], 0.99 * w_['rgb/%s/moving_mean' % scope] + 0.01 * mean,
                                   rtol=3e-2, atol=3e-3)
        np.testing.assert_allclose(out['rgb/%s/moving_variance' % scope],
                                   0.99 * w_['rgb/%s/moving_variance' % scope] + 0.01 * var, rtol=3e-2, atol=3e-3)
    # at step 0 the x2 deconv kernel is still the bilinear constant (zero off the channel diagonal), so output channels
    # U.. of first_deconvolution_conv have exactly zero gradient ONCE; both deconv kernels are dense after this step
    k = 'rgb/first_deconvolution_conv/kernel'
    assert np.array_equal(out[k][..., U:], w_[k][..., U:]) and not np.array_equal(out[k][..., :U], w_[k][..., :U])
    for scope in ('first_deconvolution_upconv', 'second_deconvolution_upconv'):
        kk = out['rgb/%s/kernel' % scope]
        assert kk.shape == w_['rgb/%s/kernel' % scope].shape and np.abs(kk[:, :, 0, 1]).max() > 0
    first = net._train_batch(data)
    for _ in range(6):
        last = net._train_batch(data)
    assert last < first
    net.fit(data, 2, output=False)
    pred = net.predict(data)
    assert pred.shape == (2, h, w)
    saved = np.load(net.export_weights())
    ref = ao.adapnet_forward(data['rgb'], {k: saved[k] for k in saved.files}, 'rgb', policy='bf16')['score']
    # A handful of large steps can collapse the net onto near-constant logits whose top two classes are a rounding error
    # apart (then the label maps agree everywhere or nowhere): compare the logits, and the labels on clear margins
    got = net.predict(data, output_attr='score')
    scale = np.abs(ref).max()
    assert np.abs(got - ref).max() < 3e-2 * scale
    top2 = np.sort(ref, -1)[..., -2:]
    clear = (top2[..., 1] - top2[..., 0]) > 6e-2 * scale
    assert np.array_equal(pred[clear], fo.argmax_last(fo.softmax(ref))[clear])


def test_adapnet_with_the_reference_tests_configuration(ops):
    """num_classes 14, num_units 20 (xview/models/test_adapnet.py:4-8): unit counts that are not lane multiples (U is
    padded to 64 zero lanes inside, the variables keep their reference shapes), inference and one training step."""
    from modular_semantic_segmentation_amd.adapnet import AdapnetEngine
    from modular_semantic_segmentation_amd.adapnet_trainer import AdapnetTrainer
    c, u, h, w = 14, 20, 32, 48
    rng = np.random.default_rng(2)
    x = rng.integers(0, 256, (2, h, w, 3)).astype(np.float32)
    labels = rng.integers(-1, c, (2, h, w)).astype(np.int32)
    w_ = ao.init_adapnet_weights('rgb', 3, u, c, seed=4, gain=1.3, blocks=SHALLOW)
    w_['rgb/block_0_1/kernel'] *= 0.02
    eng = AdapnetEngine('rgb', 3, u, c, w_, blocks=SHALLOW)
    out = eng.forward(_dev(x), want=('score', 'label'))
    ref = ao.adapnet_forward(x, w_, 'rgb', policy='bf16', blocks=SHALLOW)['score']
    score = out['score'].cpu().numpy()
    assert score.shape == (2, h, w, c) and np.abs(score - ref).max() / np.abs(ref).max() < 2e-2
    tr = AdapnetTrainer(eng, 'adam', 1e-3)
    tr.load_from_variables(w_)
    loss = tr.step(_dev(x), _dev(labels))
    ref_loss, ref_g, _ = ao.adapnet_loss_and_grads(x, labels, w_, 'rgb', c, policy='bf16', blocks=SHALLOW)
    assert abs(loss.item() - ref_loss) < 1e-2 * abs(ref_loss)
    got = tr.grads_as_variables()
    assert {k: v.shape for k, v in got.items()} == {k: v.shape for k, v in ref_g.items()}
    rel, cos = _grad_agreement(got, ref_g)
    assert rel['rgb/second_deconvolution_upconv/gamma'] < 0.05 and rel['rgb/shortcut/kernel'] < 0.2, rel
    assert min(cos.values()) > 0.8, min(cos.items(), key=lambda kv: kv[1])
    out_vars = dict(w_)
    tr.to_variables(out_vars)
    assert {k: np.shape(v) for k, v in out_vars.items()} == {k: v.shape for k, v in w_.items()}
