"""The reference's L1 / L2 functional entry points under their own names -- custom_layers.conv2d / deconv2d /
bilinear_filter_initializer (custom_layers.py:8-25,71-139), utils.cross_entropy (utils.py:43-53), simple_fcn.decoder
(simple_fcn.py:90-134) -- over the HIP kernels, against the oracle's restatement of the same TF ops on the same inputs.
Operands are rounded to bf16 before both sides see them, so what is compared is fp32 accumulation + one bf16 output
rounding (2^-8 relative) where the layer returns an Act, float32 where it returns a dense tensor."""
import os

import numpy as np
import pytest
import torch

from oracle import fcn_oracle as fo

pytestmark = pytest.mark.gpu
BF16_EPS = 2.0 ** -8


def _bf(a):
    return fo.round_bf16(np.asarray(a, np.float32))


def _nchw(x):
    return torch.from_numpy(np.ascontiguousarray(x.transpose(0, 3, 1, 2)))


def _nhwc(t):
    return t.numpy().transpose(0, 2, 3, 1)


def _close(got, ref, rel=2 * BF16_EPS, what=''):
    scale = np.abs(ref).max()
    err = np.abs(got - ref).max()
    assert err <= rel * scale + 1e-6, '%s: %.3g of max|ref| %.3g' % (what, err / scale, scale)


def _bn_vars(rng, prefix, c):
    return {prefix + '/gamma': rng.uniform(0.5, 1.5, c).astype(np.float32), prefix + '/beta': rng.normal(0, 0.2, c).astype(np.float32),
            prefix + '/moving_mean': rng.normal(0, 0.3, c).astype(np.float32),
            prefix + '/moving_variance': rng.uniform(0.5, 2.0, c).astype(np.float32)}


def _bn_ref(z, v, prefix):
    s = v[prefix + '/gamma'] / np.sqrt(v[prefix + '/moving_variance'] + 1e-3)
    return z * s + (v[prefix + '/beta'] - v[prefix + '/moving_mean'] * s)


@pytest.fixture(scope='module')
def cl():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from modular_semantic_segmentation_amd import custom_layers
    custom_layers.clear_layer_cache()
    return custom_layers


@pytest.mark.parametrize('k,cin,cout,relu', [(3, 64, 128, True), (1, 128, 64, False), (3, 128, 20, True)])
def test_conv2d_matches_tf_layers_conv2d(cl, k, cin, cout, relu):
    rng = np.random.default_rng(k * 100 + cout)
    x = _bf(rng.normal(0, 1, (2, 32, 48, cin)))
    v = {'m/l/kernel': _bf(rng.normal(0, 0.05, (k, k, cin, cout))), 'm/l/bias': rng.normal(0, 0.1, cout).astype(np.float32)}
    y = cl.conv2d(torch.from_numpy(x).cuda(), cout, [k, k], name='l', activation='relu' if relu else None, padding='same',
                  reuse=None, trainable=True, variables=v, scope='m')
    torch.cuda.synchronize()
    assert y.filters == cout and y.c == (cout + 63) // 64 * 64
    got = y.real().cpu().numpy()
    ref = _nhwc(fo.conv2d_same(_nchw(x), v['m/l/kernel'], v['m/l/bias'], relu=relu))
    _close(got[..., :cout], ref, what='conv2d %dx%d' % (k, k))
    assert not got[..., cout:].any()                          # padding lanes stay zero


def test_conv2d_first_layer_with_folded_batch_norm(cl):
    """3 input channels -> the fp32-operand first-layer kernel; batch_normalization=True, training=False: the moving
    statistics are folded into kernel and bias, the activation follows the normalisation (custom_layers.py:127-136)."""
    rng = np.random.default_rng(3)
    x = rng.integers(0, 256, (2, 32, 48, 3)).astype(np.float32)
    v = {'rgb/conv1_1/kernel': rng.normal(0, 0.02, (3, 3, 3, 64)).astype(np.float32),
         'rgb/conv1_1/bias': rng.normal(0, 0.1, 64).astype(np.float32)}
    v.update(_bn_vars(rng, 'rgb/conv1_1', 64))
    y = cl.conv2d(torch.from_numpy(x).cuda(), 64, 3, batch_normalization=True, training=False, name='conv1_1',
                  activation=torch.relu, padding='same', variables=v, scope='rgb')
    torch.cuda.synchronize()
    z = _nhwc(fo.conv2d_same(_nchw(x), v['rgb/conv1_1/kernel'], v['rgb/conv1_1/bias']))
    ref = np.maximum(_bn_ref(z, v, 'rgb/conv1_1'), 0)
    _close(y.real().cpu().numpy(), ref, what='conv1_1 + folded batch norm')


def test_conv2d_training_mode_batch_norm_updates_the_moving_statistics(cl):
    rng = np.random.default_rng(4)
    x = _bf(rng.normal(0, 1, (2, 32, 48, 64)))
    v = {'m/c/kernel': _bf(rng.normal(0, 0.05, (3, 3, 64, 64))), 'm/c/bias': rng.normal(0, 0.1, 64).astype(np.float32)}
    v.update(_bn_vars(rng, 'm/c', 64))
    before = {k_: a.copy() for k_, a in v.items()}
    y = cl.conv2d(torch.from_numpy(x).cuda(), 64, 3, batch_normalization=True, training=True, name='c', activation='relu',
                  padding='same', variables=v, scope='m')
    torch.cuda.synchronize()
    z = _bf(_nhwc(fo.conv2d_same(_nchw(x), before['m/c/kernel'], before['m/c/bias'])))   # the kernel's z is stored as bf16
    mean, var = z.reshape(-1, 64).mean(0, dtype=np.float64), z.reshape(-1, 64).var(0, dtype=np.float64)
    ref = np.maximum((z - mean) / np.sqrt(var + 1e-3) * before['m/c/gamma'] + before['m/c/beta'], 0)
    _close(y.real().cpu().numpy(), ref, rel=4 * BF16_EPS, what='training-mode batch norm')
    np.testing.assert_allclose(v['m/c/moving_mean'], 0.99 * before['m/c/moving_mean'] + 0.01 * mean, rtol=0, atol=2e-4)
    assert np.abs(v['m/c/moving_variance'] - before['m/c/moving_variance']).max() > 1e-4      # updated (UPDATE_OPS)


@pytest.mark.parametrize('stride,bn', [(2, False), (8, False), (8, True), (2, True)])
def test_deconv2d_bilinear_matches_conv2d_transpose(cl, stride, bn):
    rng = np.random.default_rng(10 + stride)
    k, c = 2 * stride, 64
    x = _bf(rng.normal(0.2, 1, (2, 6, 9, c)))
    v = _bn_vars(rng, 'm/up', c) if bn else {}
    y = cl.deconv2d(torch.from_numpy(x).cuda(), c, [k, k], strides=[stride, stride], padding='same', activation='relu',
                    name='up', trainable=False, batch_normalization=bn, training=False, variables=v, scope='m')
    torch.cuda.synchronize()
    z = _nhwc(fo.deconv_same(_nchw(x), fo.bilinear_kernel(k, c), stride))
    if bn:
        z = _bn_ref(_bf(z), v, 'm/up')      # the x8 map is stored as bf16 before the affine pass
    _close(y.real().cpu().numpy(), np.maximum(z, 0), rel=3 * BF16_EPS, what='bilinear x%d' % stride)
    assert (y.n, y.h, y.w, y.c) == (2, 6 * stride, 9 * stride, c)


def test_deconv2d_with_a_trained_dense_kernel(cl):
    rng = np.random.default_rng(12)
    c = 64
    x = _bf(rng.normal(0, 1, (1, 8, 12, c)))
    v = {'m/up/kernel': _bf(rng.normal(0, 0.05, (4, 4, c, c)))}
    y = cl.deconv2d(torch.from_numpy(x).cuda(), c, 4, strides=2, padding='same', activation=None, name='up',
                    batch_normalization=False, variables=v, scope='m')
    torch.cuda.synchronize()
    ref = _nhwc(fo.deconv_same(_nchw(x), v['m/up/kernel'], 2))
    _close(y.real().cpu().numpy(), ref, what='dense x2 deconv')


def test_bilinear_filter_initializer_is_the_reference_constant(cl, golden_dir):
    g = np.load(os.path.join(golden_dir, 'bilinear_kernels.npz'))
    init = cl.bilinear_filter_initializer([4, 4, 3, 7])
    np.testing.assert_array_equal(init([4, 4, 3, 7]), g['k4_rect'].astype(np.float32))
    np.testing.assert_array_equal(cl.bilinear_filter_initializer((16, 16, 2, 5))(), g['k16_rect'].astype(np.float32))
    with pytest.raises(ValueError):
        init([4, 4, 3, 3])                                     # verify_shape=True


@pytest.mark.parametrize('labels_as', ['int', 'onehot'])
def test_cross_entropy_matches_utils_cross_entropy(cl, labels_as):
    from modular_semantic_segmentation_amd.utils import cross_entropy
    rng = np.random.default_rng(20)
    c = 12
    score = rng.normal(0, 2, (2, 16, 24, c)).astype(np.float32)
    lab = rng.integers(-1, c, (2, 16, 24)).astype(np.int32)
    logp = fo.log_softmax(score)
    want, wgrad = fo.cross_entropy(logp, lab, c)
    if labels_as == 'int':
        labels = torch.from_numpy(lab).cuda()
    else:
        hot = np.zeros(score.shape, np.float32)
        idx = np.nonzero(lab >= 0)
        hot[idx + (lab[idx],)] = 1.0
        labels = torch.from_numpy(hot).cuda()
    loss, grad = cross_entropy(torch.from_numpy(logp).cuda(), labels, return_gradient=True)
    torch.cuda.synchronize()
    assert abs(float(loss) - want) <= 1e-5 * abs(want)
    np.testing.assert_allclose(grad.cpu().numpy(), wgrad, rtol=0, atol=1e-6 * np.abs(wgrad).max() + 1e-9)
    # the scores themselves give the same loss (log_softmax is idempotent), and no labelled pixel gives 0 (the 1e-20 guard)
    assert abs(float(cross_entropy(torch.from_numpy(score).cuda(), labels)) - want) <= 1e-5 * abs(want)
    none = torch.full((2, 16, 24), -1, dtype=torch.int32).cuda()
    assert float(cross_entropy(torch.from_numpy(logp).cuda(), none)) == 0.0


@pytest.mark.parametrize('batchnorm,units', [(True, 64), (False, 64), (True, 20)])
def test_decoder_matches_simple_fcn_decoder(cl, batchnorm, units):
    """simple_fcn.decoder: x8 bilinear deconv [+ BN] + relu, then the 1x1 score conv [+ BN] (no activation) -- against the
    same two TF layers restated with the oracle's ops; units = 20 runs on 64 padded lanes."""
    from modular_semantic_segmentation_amd.simple_fcn import decoder
    rng = np.random.default_rng(30 + units)
    c = 12
    feat = np.maximum(_bf(rng.normal(0.3, 1, (2, 4, 6, units))), 0)
    v = {'rgb/score/kernel': _bf(rng.normal(0, 0.2, (1, 1, units, c))), 'rgb/score/bias': rng.normal(0, 0.1, c).astype(np.float32)}
    if batchnorm:
        v.update(_bn_vars(rng, 'rgb/upscore', units))
        v.update(_bn_vars(rng, 'rgb/score', c))
    x = np.zeros((2, 4, 6, 64), np.float32)
    x[..., :units] = feat
    act = cl._as_act(torch.from_numpy(x).cuda())
    act.filters = units
    out = decoder(act, 'rgb', units, c, is_training=False, batchnorm=batchnorm, variables=v)
    torch.cuda.synchronize()
    up = _bf(_nhwc(fo.deconv_same(_nchw(feat), fo.bilinear_kernel(16, units), 8)))
    if batchnorm:
        up = _bn_ref(up, v, 'rgb/upscore')
    up = _bf(np.maximum(up, 0))
    score = _nhwc(fo.conv2d_same(_nchw(up), v['rgb/score/kernel'], v['rgb/score/bias']))
    if batchnorm:
        score = _bn_ref(score, v, 'rgb/score')
    got_up = out['upscore'].real().cpu().numpy()
    _close(got_up[..., :units], up, rel=3 * BF16_EPS, what='upscore')
    assert out['score'].shape == (2, 32, 48, c) and out['score'].dtype == torch.float32
    _close(out['score'].cpu().numpy(), score, rel=4 * BF16_EPS, what='score')
