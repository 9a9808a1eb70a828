"""conv_dtype='fp32' ("exact" mode, csrc/exact_f32.hip + fcn_exact.FcnEngineF32): the FCN graph of
xview/models/simple_fcn.py:10-170 in plain float32 -- the reference graph's own arithmetic type -- through the C ABI
(xv_conv2d_f32, xv_maxpool2x2_f32, xv_upsample2x_f32, xv_score_lowres_f32, xv_decoder_head_from_scores) against the fp32
oracle.  Only fp32 summation order separates the two: logits within 1e-5 of the logit scale, label maps identical wherever
the oracle's top-2 margin exceeds that.  (On TRAINED weights at 768x384: tests/test_zz_accuracy_gpu.py.)"""
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


@pytest.mark.parametrize('n,h,w,cin,cout,k,relu', [(2, 16, 24, 3, 64, 3, True), (1, 13, 9, 64, 80, 3, False),
                                                   (1, 8, 8, 512, 64, 1, True), (2, 24, 40, 128, 256, 3, True)])
def test_conv2d_f32_against_oracle(lib, n, h, w, cin, cout, k, relu):
    rng = np.random.default_rng(n * h + cin)
    xi = rng.integers(-3, 4, (n, h, w, cin)).astype(np.float32)
    wi = rng.integers(-2, 3, (k, k, cin, cout)).astype(np.float32)
    b = rng.integers(-3, 4, cout).astype(np.float32)
    for x, wt, exact in ((xi, wi, True), (rng.standard_normal(xi.shape).astype(np.float32),
                                          (rng.standard_normal(wi.shape) / np.sqrt(k * k * cin)).astype(np.float32), False)):
        xd, wd, bd = (torch.from_numpy(a).cuda() for a in (x, wt, b))
        y = torch.full((n, h, w, cout), float('nan'), device='cuda')
        assert lib.xv_conv2d_f32(_p(xd), n, h, w, cin, _p(wd), _p(bd), k, cout, int(relu), _p(y), None) == 0
        torch.cuda.synchronize()
        ref = fo.conv2d_same(torch.from_numpy(x).permute(0, 3, 1, 2).contiguous(), wt, b, relu=relu).permute(0, 2, 3, 1).numpy()
        got = y.cpu().numpy()
        if exact:
            assert np.array_equal(got, ref)                      # sums of small integers: exact in any order
        else:
            assert np.abs(got - ref).max() <= 2e-6 * np.abs(ref).max() + 1e-6


def test_pool_and_upsample_f32(lib):
    rng = np.random.default_rng(3)
    x = rng.standard_normal((2, 8, 12, 64)).astype(np.float32)
    xd = torch.from_numpy(x).cuda()
    q = torch.empty((2, 4, 6, 64), device='cuda')
    assert lib.xv_maxpool2x2_f32(_p(xd), 2, 8, 12, 64, _p(q), None) == 0
    ref = fo.maxpool2(torch.from_numpy(x).permute(0, 3, 1, 2)).permute(0, 2, 3, 1).numpy()
    assert np.array_equal(q.cpu().numpy(), ref)
    res = rng.standard_normal((2, 16, 24, 64)).astype(np.float32)
    y = torch.empty((2, 16, 24, 64), device='cuda')
    assert lib.xv_upsample2x_f32(_p(xd), 2, 8, 12, 64, _p(torch.from_numpy(res).cuda()), _p(y), None) == 0
    torch.cuda.synchronize()
    up = fo.deconv_same(torch.from_numpy(x).permute(0, 3, 1, 2).contiguous(), fo.bilinear_kernel(4, 64), 2)
    want = torch.relu(up).permute(0, 2, 3, 1).numpy() + res
    assert np.abs(y.cpu().numpy() - want).max() < 1e-6


@pytest.mark.parametrize('cin', [3, 1])
def test_exact_engine_matches_fp32_oracle(lib, cin):
    """The whole expert at 64x96 and 128x160 on random-init weights: logits within 1e-5 of the logit scale of the fp32
    oracle (fp32 summation order only), labels identical on every pixel whose top-2 margin exceeds twice that."""
    from modular_semantic_segmentation_amd.fcn_exact import FcnEngineF32
    prefix = 'rgb' if cin == 3 else 'depth'
    w = fo.init_fcn_weights(prefix, cin, 64, 12, seed=1, bias_scale=0.02)
    w['%s/conv1_1/kernel' % prefix] *= 0.02 if cin == 3 else 0.02 / 256
    for k in w:
        if k.endswith('/kernel') and 'upscore' not in k and 'conv1_1' not in k:
            w[k] *= 1.6
    eng = FcnEngineF32(prefix, cin, 64, 12, w)
    rng = np.random.default_rng(0)
    for h, wd in ((64, 96), (128, 160)):
        x = rng.integers(0, 256 if cin == 3 else 65536, (2, h, wd, cin)).astype(np.float32)
        out = eng.forward(torch.from_numpy(x).cuda(), want=('score', 'prob', 'label'))
        torch.cuda.synchronize()
        ref = fo.fcn_forward(x, w, prefix, 'fp32', keep=['conv3_3', 'fused', 'score'])
        for name in ('conv3_3', 'fused'):
            got = out['layers'][name].cpu().numpy()
            assert np.abs(got - ref[name]).max() <= 1e-5 * np.abs(ref[name]).max(), name
        score = out['score'].cpu().numpy()
        scale = np.abs(ref['score']).max()
        err = np.abs(score - ref['score']).max() / scale
        print('fp32 engine vs fp32 oracle at %dx%d: max logit error %.2e of the scale' % (wd, h, err))
        assert err < 1e-5
        lab = out['label'].cpu().numpy()
        assert np.array_equal(lab, fo.argmax_last(fo.softmax(score)))
        top2 = np.sort(ref['score'], -1)[..., -2:]
        clear = (top2[..., 1] - top2[..., 0]) > 2e-5 * scale
        assert clear.mean() > 0.99 and np.array_equal(lab[clear], fo.argmax_last(fo.softmax(ref['score']))[clear])
        assert np.allclose(out['prob'].cpu().numpy(), fo.softmax(score), atol=1e-6)


@pytest.mark.parametrize('cin', [3, 1])
def test_exact_engine_full_size_labels_equal_fp32_oracle(lib, cin):
    """BASELINE's headline size, 768x384, on RANDOM-init weights (no training loop: the exact-mode evidence must not hang on
    the statistical test of tests/test_zz_accuracy_gpu.py): conv_dtype='fp32' against the fp32 oracle -- logits within 1e-5
    of the logit scale, argmax label map IDENTICAL on every pixel whose fp32 top-2 margin exceeds twice that bound (which is
    all but a handful of the 294 912), and the differing pixels counted and printed."""
    from modular_semantic_segmentation_amd.fcn_exact import FcnEngineF32
    prefix = 'rgb' if cin == 3 else 'depth'
    w = fo.init_fcn_weights(prefix, cin, 64, 12, seed=5, bias_scale=0.02)
    w['%s/conv1_1/kernel' % prefix] *= 0.02 if cin == 3 else 0.02 / 256
    for k in w:
        if k.endswith('/kernel') and 'upscore' not in k and 'conv1_1' not in k:
            w[k] *= 1.6
    eng = FcnEngineF32(prefix, cin, 64, 12, w)
    rng = np.random.default_rng(11)
    x = rng.integers(0, 256 if cin == 3 else 65536, (1, 384, 768, cin)).astype(np.float32)
    out = eng.forward(torch.from_numpy(x).cuda(), want=('score', 'label'))
    torch.cuda.synchronize()
    ref = fo.fcn_forward(x, w, prefix, 'fp32')['score']
    score = out['score'].cpu().numpy()
    scale = np.abs(ref).max()
    err = np.abs(score - ref).max() / scale
    lab, ref_lab = out['label'].cpu().numpy(), fo.argmax_last(fo.softmax(ref))
    top2 = np.sort(ref, -1)[..., -2:]
    clear = (top2[..., 1] - top2[..., 0]) > 2e-5 * scale
    print('fp32 engine vs fp32 oracle at 768x384 (%s): max logit error %.2e of the scale, %d of %d labels differ, %d pixels '
          'inside the 2e-5 margin' % (prefix, err, int((lab != ref_lab).sum()), lab.size, int((~clear).sum())))
    assert err < 1e-5
    assert clear.mean() > 0.999 and np.array_equal(lab[clear], ref_lab[clear])
    assert (lab != ref_lab).sum() <= (~clear).sum()


def test_exact_mode_through_the_model_api(lib, golden_dir):
    import os
    from modular_semantic_segmentation_amd import get_model
    g = np.load(os.path.join(golden_dir, 'notebook_868.npz'))
    desc = ({'rgb': 'float32', 'depth': 'float32', 'labels': 'int32'},
            {'rgb': (None, None, 3), 'depth': (None, None, 1), 'labels': (None, None)}, 12)
    cfg = dict(data_description=desc, confusion_matrices={'rgb': g['cm_rgb'], 'depth': g['cm_depth']}, num_units=64,
               prefixes={'rgb': 'rgb', 'depth': 'depth'}, num_channels={'rgb': 3, 'depth': 1}, expert_model='fcn',
               class_prior='data', batchsize=2, seed=3)
    rng = np.random.default_rng(1)
    data = {'rgb': rng.integers(0, 256, (2, 64, 96, 3)).astype(np.float32),
            'depth': rng.integers(0, 65536, (2, 64, 96, 1)).astype(np.float32)}
    preds = {}
    for dt in ('bf16', 'fp32'):
        net = get_model('bayes_fusion')(conv_dtype=dt, **cfg)
        net.variables['depth/conv1_1/kernel'] = net.variables['depth/conv1_1/kernel'] / 256.0
        net._variables_changed()
        preds[dt] = net.predict(data)
        score = net.predict(data, output_attr='fused_score')
        assert np.array_equal(preds[dt], np.argmax(score, -1))
    assert preds['fp32'].shape == (2, 64, 96) and (preds['fp32'] == preds['bf16']).mean() > 0.9
    with pytest.raises(UserWarning):
        get_model('fcn')('rgb', desc, 'rgb', num_units=64, batch_normalization=False, batchsize=2, conv_dtype='fp32')._ensure_trainer()
