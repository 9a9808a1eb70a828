"""BASELINE.json's full sizes (768x384, 1024x512, 2048x1024) through size-independent properties,
plus one full-size comparison with the oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import fcn_oracle as fo
from oracle import fusion_oracle as fu

C, U = 12, 64


@pytest.fixture(scope='module')
def ops():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from modular_semantic_segmentation_amd import ops as _ops
    return _ops


def _weights(prefix, cin, seed, scale_first):
    w = fo.init_fcn_weights(prefix, cin, U, C, seed=seed, bias_scale=0.02)
    w['%s/conv1_1/kernel' % prefix] *= scale_first
    for k in w:
        if k.endswith('/kernel') and 'upscore' not in k and 'conv1_1' not in k:
            w[k] *= 1.6
    return w


@pytest.mark.parametrize('h,w,cin,cout', [(384, 768, 64, 64), (256, 512, 128, 128), (128, 256, 512, 512)])
def test_conv_linearity_and_shift_equivariance_full_size(ops, h, w, cin, cout):
    """conv(2x) == 2 conv(x) and conv(shift(x)) == shift(conv(x)) bit for bit (bias 0, no relu): the
    per-output accumulation order does not depend on the tile an output falls into."""
    g = torch.Generator(device='cuda').manual_seed(0)
    x = torch.randn((1, h, w, cin), device='cuda', generator=g)
    wt = torch.randn((3, 3, cin, cout), device='cuda', generator=g) * (1.0 / (9 * cin) ** 0.5)
    wp = ops.pack_conv_weights(wt)
    b = torch.zeros(cout, device='cuda')
    y1, _ = ops.conv2d_fwd(ops.Act.from_dense(x), wp, b, 3, relu=False)
    y2, _ = ops.conv2d_fwd(ops.Act.from_dense(2 * x), wp, b, 3, relu=False)
    torch.cuda.synchronize()
    assert torch.equal(y2.interior().float(), 2 * y1.interior().float())
    # shift by (24, 40): not a multiple of any tile size
    xs = torch.zeros_like(x)
    xs[:, 24:, 40:] = x[:, :-24, :-40]
    y3, _ = ops.conv2d_fwd(ops.Act.from_dense(xs), wp, b, 3, relu=False)
    torch.cuda.synchronize()
    # away from the borders of both images (first shifted rows/cols and the last row/col see zero padding)
    a = y3.interior()[:, 26:-1, 42:-1].float()
    bref = y1.interior()[:, 2:-25, 2:-41].float()
    assert torch.equal(a, bref)


def test_fcn_full_size_against_oracle(ops):
    """One 768x384 RGB image end to end against the bf16-policy oracle."""
    from modular_semantic_segmentation_amd.fcn import FcnEngine
    w = _weights('rgb', 3, 1, 0.02)
    rng = np.random.default_rng(0)
    x = rng.integers(0, 256, (1, 384, 768, 3)).astype(np.float32)
    eng = FcnEngine('rgb', 3, U, C, w)
    out = eng.forward(torch.from_numpy(x).cuda(), want=('score', 'label'))
    torch.cuda.synchronize()
    ref = fo.fcn_forward(x, w, 'rgb', 'bf16')['score']
    got = out['score'].cpu().numpy()
    scale = np.abs(ref).max()
    print('full-size logits vs the bf16-policy oracle: max error %.5f of the logit scale' % (np.abs(got - ref).max() / scale))
    from tolerances import LOGIT_TOL_VS_POLICY
    assert np.abs(got - ref).max() / scale < LOGIT_TOL_VS_POLICY['bf16']        # measured 1.2e-2 (13 conv layers of bf16 storage)
    lab = out['label'].cpu().numpy()
    assert np.array_equal(lab, fo.argmax_last(fo.softmax(got)))
    ref_lab = fo.argmax_last(fo.softmax(ref))
    top2 = np.sort(ref, -1)[..., -2:]
    clear = (top2[..., 1] - top2[..., 0]) > 4e-2 * scale
    assert np.array_equal(lab[clear], ref_lab[clear])
    assert (lab == ref_lab).mean() > 0.97


@pytest.mark.parametrize('h,w', [(512, 1024), (1024, 2048)])
def test_bayes_fusion_model_large_images(ops, golden_dir, h, w):
    """Configs 3/4 sizes: batch independence (an image gives the same labels alone or in a batch),
    LUT path == per-pixel path, and conservation laws of the confusion matrix."""
    from modular_semantic_segmentation_amd import get_model
    g = np.load(os.path.join(golden_dir, 'notebook_868.npz'))
    cms = {'rgb': g['cm_rgb'], 'depth': g['cm_depth']}
    desc = ({'rgb': 'float32', 'depth': 'float32', 'labels': 'int32'},
            {'rgb': (None, None, 3), 'depth': (None, None, 1), 'labels': (None, None)}, C)
    net = get_model('bayes_fusion')(data_description=desc, confusion_matrices=cms, num_units=U,
                                    prefixes={'rgb': 'rgb', 'depth': 'depth'}, num_channels={'rgb': 3, 'depth': 1},
                                    expert_model='fcn', class_prior='data', batchsize=2, seed=3)
    net.variables['depth/conv1_1/kernel'] = net.variables['depth/conv1_1/kernel'] / 256.0
    net._variables_changed()
    rng = np.random.default_rng(1)
    data = {'rgb': rng.integers(0, 256, (2, h, w, 3)).astype(np.float32),
            'depth': rng.integers(0, 65536, (2, h, w, 1)).astype(np.float32),
            'labels': rng.integers(-1, C, (2, h, w)).astype(np.int32)}
    both = net.predict(data)                         # default path: fused two-expert head
    assert both.shape == (2, h, w) and both.dtype == np.int64
    score = net.predict(data, output_attr='fused_score')          # unfused path, expert label maps materialised
    assert np.array_equal(both, np.argmax(score, -1))             # fused head == head + fusion kernels, bit for bit
    la = net.expert_outputs['rgb']['classification'].clone()
    lb = net.expert_outputs['depth']['classification'].clone()
    single = net.predict({k: v[1:2] for k, v in data.items()})
    assert np.array_equal(single[0], both[1])
    lut = ops.bayes_fuse_lut(la, lb, net.decision_matrix).cpu().numpy()
    top2 = np.sort(score, -1)[..., -2:]
    clear = (top2[..., 1] - top2[..., 0]) > 1e-4
    assert np.array_equal(lut[clear], both[clear])
    measures, cm = net.score(data)
    assert cm.sum() == (data['labels'] >= 0).sum()
    assert np.array_equal(cm.sum(1), np.bincount(data['labels'][data['labels'] >= 0], minlength=C))
    assert np.array_equal(cm, fu.confusion_matrix(data['labels'], both, C).astype(np.float64))
