"""conv_dtype='fp32' on the matrix cores (csrc/exact_f32.hip, conv_f32_mfma_kernel on v_mfma_f32_32x32x2_f32): the new entry
points xv_conv2d_f32_pool / xv_conv2d_f32_scalar through the C ABI against the fp32 oracle (conv2d 'same' + bias + relu,
max_pooling2d 2x2: simple_fcn.py:39-67) -- exact on small integers (any summation order gives the same float), within fp32
summation order on random operands, on shapes that exercise every mask of the kernel (partial tiles in both directions,
channel counts that are not multiples of 4 / 16 / 64, 1x1), pooled-only and full-only outputs, and bit-identical results
for an image alone and inside a batch.  tests/test_exact_f32_gpu.py (unchanged since round 4) runs the same kernel through
xv_conv2d_f32 and the whole engine."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import fcn_oracle as fo


@pytest.fixture(scope='module')
def lib():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from modular_semantic_segmentation_amd import _lib
    return _lib.lib()


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _oracle(x, w, b, relu):
    y = fo.conv2d_same(torch.from_numpy(x).permute(0, 3, 1, 2).contiguous(), w, b, relu=relu)
    return y.permute(0, 2, 3, 1).numpy(), fo.maxpool2(y).permute(0, 2, 3, 1).numpy()


SHAPES = [(2, 16, 64, 16, 64, 3, True),      # whole tiles, one chunk
          (1, 10, 38, 20, 72, 3, True),      # partial tiles both ways, cin not a multiple of 16, cout not of 64
          (2, 6, 4, 3, 5, 3, False),         # tiny map, scalar input path (cin % 4 != 0), cout % 4 != 0 (scalar stores)
          (1, 24, 48, 64, 128, 3, True),     # conv5-like map: 3 x 1.5 tiles, four chunks, two channel blocks
          (1, 8, 40, 1, 64, 3, True),        # depth conv1_1: one input channel
          (2, 12, 20, 40, 24, 1, True),      # 1x1
          (1, 48, 96, 256, 64, 1, False)]    # score_conv-like 1x1 over 16 chunks


@pytest.mark.parametrize('n,h,w,cin,cout,k,relu', SHAPES)
def test_conv2d_f32_pool_against_oracle(lib, n, h, w, cin, cout, k, relu):
    rng = np.random.default_rng(7 * n + h + cin + cout)
    xi = rng.integers(-3, 4, (n, h, w, cin)).astype(np.float32)
    wi = rng.integers(-2, 3, (k, k, cin, cout)).astype(np.float32)
    b = rng.integers(-3, 4, cout).astype(np.float32)
    xr = rng.standard_normal(xi.shape).astype(np.float32)
    wr = (rng.standard_normal(wi.shape) / np.sqrt(k * k * cin)).astype(np.float32)
    for x, wt, exact in ((xi, wi, True), (xr, wr, False)):
        xd, wd, bd = (torch.from_numpy(a).cuda() for a in (x, wt, b))
        ref, ref_pool = _oracle(x, wt, b, relu)
        for want_y, want_p in ((True, True), (True, False), (False, True)):
            y = torch.full((n, h, w, cout), float('nan'), device='cuda')
            q = torch.full((n, h // 2, w // 2, cout), float('nan'), device='cuda')
            rc = lib.xv_conv2d_f32_pool(_p(xd), n, h, w, cin, _p(wd), _p(bd), k, cout, int(relu), _p(y if want_y else None),
                                        _p(q if want_p else None), None)
            assert rc == 0
            torch.cuda.synchronize()
            for got, want, on in ((y.cpu().numpy(), ref, want_y), (q.cpu().numpy(), ref_pool, want_p)):
                if not on:
                    assert np.isnan(got).all()                       # an output that was not asked for is not touched
                elif exact:
                    assert np.array_equal(got, want)
                else:
                    assert np.abs(got - want).max() <= 2e-6 * np.abs(want).max() + 1e-6
        # the round-4 vector-ALU kernel (bench A/B baseline) computes the same function
        ys = torch.empty((n, h, w, cout), device='cuda')
        assert lib.xv_conv2d_f32_scalar(_p(xd), n, h, w, cin, _p(wd), _p(bd), k, cout, int(relu), _p(ys), None) == 0
        torch.cuda.synchronize()
        if exact:
            assert np.array_equal(ys.cpu().numpy(), ref)
        else:
            assert np.abs(ys.cpu().numpy() - ref).max() <= 2e-6 * np.abs(ref).max() + 1e-6


def test_conv2d_f32_argument_checks(lib):
    x = torch.zeros((1, 7, 8, 4), device='cuda')
    w = torch.zeros((3, 3, 4, 8), device='cuda')
    b = torch.zeros(8, device='cuda')
    y = torch.zeros((1, 7, 8, 8), device='cuda')
    q = torch.zeros((1, 3, 4, 8), device='cuda')
    assert lib.xv_conv2d_f32_pool(_p(x), 1, 7, 8, 4, _p(w), _p(b), 3, 8, 1, _p(y), _p(q), None) == -2     # odd height, pooled
    assert lib.xv_conv2d_f32_pool(_p(x), 1, 7, 8, 4, _p(w), _p(b), 3, 8, 1, None, None, None) == -1        # no output
    assert lib.xv_conv2d_f32_pool(_p(x), 1, 7, 8, 4, _p(w), _p(b), 5, 8, 1, _p(y), None, None) == -2       # kernel size
    assert lib.xv_conv2d_f32_pool(_p(x), 1, 7, 8, 4, _p(w), _p(b), 3, 8, 1, _p(y), None, None) == 0


def test_conv2d_f32_is_batch_independent_bit_for_bit(lib):
    """An output value is one fixed fmaf chain (chunk, tap, channel step): the same bits for an image alone and in a batch."""
    rng = np.random.default_rng(5)
    n, h, w, cin, cout = 3, 24, 40, 48, 96
    x = torch.from_numpy(rng.standard_normal((n, h, w, cin)).astype(np.float32)).cuda()
    wt = torch.from_numpy((rng.standard_normal((3, 3, cin, cout)) / 20).astype(np.float32)).cuda()
    b = torch.from_numpy(rng.standard_normal(cout).astype(np.float32)).cuda()
    y = torch.empty((n, h, w, cout), device='cuda')
    q = torch.empty((n, h // 2, w // 2, cout), device='cuda')
    assert lib.xv_conv2d_f32_pool(_p(x), n, h, w, cin, _p(wt), _p(b), 3, cout, 1, _p(y), _p(q), None) == 0
    for i in range(n):
        xi = x[i:i + 1].contiguous()
        yi = torch.empty((1, h, w, cout), device='cuda')
        qi = torch.empty((1, h // 2, w // 2, cout), device='cuda')
        assert lib.xv_conv2d_f32_pool(_p(xi), 1, h, w, cin, _p(wt), _p(b), 3, cout, 1, _p(yi), _p(qi), None) == 0
        torch.cuda.synchronize()
        assert torch.equal(yi[0], y[i]) and torch.equal(qi[0], q[i])
    # and the pooled map is the 2x2 max of the full map, exactly
    assert torch.equal(q, torch.nn.functional.max_pool2d(y.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1))


def test_exact_engine_with_batch_norm_after_every_layer(lib):
    """The reference's shipped `batch_normalization: true` in the label-exact mode: conv batch norms folded into the float32
    weights, the two deconv batch norms as per-channel affines (xv_upsample2x_affine_f32, the un-commuted float32 head
    xv_decoder_head_affine_f32) -- logits within 1e-5 of the scale of the fp32 oracle's, labels identical outside twice
    that margin, probabilities the softmax of the scores."""
    from modular_semantic_segmentation_amd.fcn import init_variables
    from modular_semantic_segmentation_amd.fcn_exact import FcnEngineF32
    rng = np.random.default_rng(3)
    w = init_variables('rgb', 3, 64, 12, batch_normalization=True, seed=2)
    w['rgb/conv1_1/kernel'] *= 0.02
    for k in list(w):
        if k.endswith('/gamma'):
            w[k] = rng.uniform(0.7, 1.3, w[k].shape).astype(np.float32)
        elif k.endswith('/beta'):
            w[k] = (0.2 * rng.standard_normal(w[k].shape)).astype(np.float32)
        elif k.endswith('/moving_mean'):
            w[k] = (0.1 * rng.standard_normal(w[k].shape)).astype(np.float32)
        elif k.endswith('/moving_variance'):
            w[k] = rng.uniform(0.5, 1.5, w[k].shape).astype(np.float32)
        elif k.endswith('/kernel') and 'upscore' not in k and 'conv1_1' not in k:
            w[k] = w[k] * 1.6
    eng = FcnEngineF32('rgb', 3, 64, 12, w)
    assert not eng.commuted_head()
    x = rng.integers(0, 256, (2, 64, 96, 3)).astype(np.float32)
    out = eng.forward(torch.from_numpy(x).cuda(), want=('score', 'prob', 'label'))
    torch.cuda.synchronize()
    ref = fo.fcn_forward(x, w, 'rgb', 'fp32', keep=['fused', 'score'])
    got = out['layers']['fused'].cpu().numpy()
    assert np.abs(got - ref['fused']).max() <= 1e-5 * np.abs(ref['fused']).max()
    score = out['score'].cpu().numpy()
    scale = np.abs(ref['score']).max()
    err = np.abs(score - ref['score']).max() / scale
    print('fp32 engine with batch norm vs fp32 oracle: max logit error %.2e of the scale' % err)
    assert err < 1e-5
    lab = out['label'].cpu().numpy()
    assert np.array_equal(lab, fo.argmax_last(fo.softmax(score)))
    top2 = np.sort(ref['score'], -1)[..., -2:]
    clear = (top2[..., 1] - top2[..., 0]) > 2e-5 * scale
    assert clear.mean() > 0.99 and np.array_equal(lab[clear], fo.argmax_last(fo.softmax(ref['score']))[clear])
    assert np.allclose(out['prob'].cpu().numpy(), fo.softmax(score), atol=1e-6)
    # label only (what predict() asks for)
    only = eng.forward(torch.from_numpy(x).cuda())
    assert torch.equal(only['label'], out['label'])
    # ... and through the model API (get_model('fcn'), batch_normalization: true, conv_dtype: 'fp32')
    import os
    import tempfile
    from modular_semantic_segmentation_amd import get_model
    desc = ({'rgb': 'float32', 'labels': 'int32'}, {'rgb': (None, None, 3), 'labels': (None, None)}, 12)
    net = get_model('fcn')('rgb', desc, 'rgb', num_units=64, batch_normalization=True, batchsize=2, conv_dtype='fp32')
    with tempfile.TemporaryDirectory() as d:
        np.savez(os.path.join(d, 'w.npz'), **w)
        net.import_weights(os.path.join(d, 'w.npz'), warnings=False)
    assert np.array_equal(np.asarray(net.predict({'rgb': x})), lab)
