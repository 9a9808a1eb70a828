"""BASELINE.json configs[3]: DirichletFusion fit + predict on full-size images (1024x512 and 2048x1024) through the
HIP path, against fusion_oracle fed the SAME expert probabilities; and the data-parallel form -- every rank measures
its own images, one all-reduce of the sufficient statistics, every rank fits -- against the single-process fit.

Reference: xview/models/dirichlet_mix.py:140-168 (statistics graph), :175-205 (accumulation), :207-257 (fit),
:96-138 (fused prediction)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

from oracle import fusion_oracle as fu

C, U = 12, 64
CFG = dict(modalities=['rgb', 'depth'], num_channels={'rgb': 3, 'depth': 1}, num_units=U, expert_model='fcn',
           class_prior='data', sigma=1.0, delta=1e-2, beta=1e-2, batchsize=1, seed=11)


def _desc():
    return ({'rgb': 'float32', 'depth': 'float32', 'labels': 'int32'},
            {'rgb': (None, None, 3), 'depth': (None, None, 1), 'labels': (None, None)}, C)


def _data(h, w, n=2):
    """Blocky label maps (every class present, some pixels unlabelled) and images whose colour / depth follow the
    label, so the experts' posteriors differ between classes and the fit has something to find."""
    rng = np.random.default_rng(h + 7)
    coarse = rng.integers(-1, C, (n, h // 32, w // 32))
    labels = np.repeat(np.repeat(coarse, 32, axis=1), 32, axis=2).astype(np.int32)
    palette = rng.integers(0, 256, (C + 1, 3)).astype(np.float32)
    dpal = rng.integers(0, 65536, (C + 1, 1)).astype(np.float32)
    rgb = np.clip(palette[labels] + rng.normal(0, 20, (n, h, w, 3)), 0, 255).astype(np.float32)
    depth = np.clip(dpal[labels] + rng.normal(0, 3000, (n, h, w, 1)), 0, 65535).astype(np.float32)
    return {'rgb': rgb, 'depth': depth, 'labels': labels}


def _make(device='cuda'):
    from modular_semantic_segmentation_amd import get_model
    net = get_model('dirichlet_fusion')(data_description=_desc(), device=device, **CFG)
    # a trained depth expert absorbs the raw uint16 range in conv1_1; random initialisers need the scale
    net.variables['depth/conv1_1/kernel'] = net.variables['depth/conv1_1/kernel'] / 256.0
    net.variables['rgb/conv1_1/kernel'] = net.variables['rgb/conv1_1/kernel'] * 0.5
    net._variables_changed()
    return net


def _expert_probs(net, data, i):
    """The experts' softmax outputs for image i, as the HIP path computes them (the oracle is fed these)."""
    out = {}
    for m in net.modalities:
        x = torch.from_numpy(data[m][i:i + 1]).to(net.device)
        out[m] = net.experts[m].forward(x, want=('prob',))['prob'].cpu().numpy()
    torch.cuda.synchronize()
    return out


def _oracle_fit(S, counts):
    """dirichlet_mix.py:207-257 on oracle statistics, through the golden-pinned Newton fitter
    (tests/test_oracle_golden.py pins dirichlet_fit.find_dirichlet_priors to the reference's to 1e-15)."""
    from modular_semantic_segmentation_amd.dirichlet_fit import find_dirichlet_priors
    params = {}
    for m in S:
        A = np.ones((C, C))
        for c in range(C):
            if counts[c] == 0:
                continue
            ss = S[m][c] / counts[c]
            neg = (S[m].sum(0) - S[m][c]) / (counts.sum() - counts[c])
            A[:, c] = find_dirichlet_priors(ss, neg, np.ones(C), max_iter=10000, delta=CFG['delta'], beta=CFG['beta'])
        params[m] = A
    return params


@pytest.mark.parametrize('h,w', [(512, 1024), (1024, 2048)])
def test_dirichlet_fit_and_predict_full_size(h, w):
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    data = _data(h, w)
    net = _make()
    params = net.fit(data)
    # ---- sufficient statistics: HIP reduction vs the oracle on the same probabilities --------------------------
    S_ref = {m: np.zeros((C, C)) for m in net.modalities}
    counts_ref = np.zeros(C, np.int64)
    probs = []
    for i in range(len(data['labels'])):
        p = _expert_probs(net, data, i)
        probs.append(p)
        for m in net.modalities:
            s, cnt = fu.sufficient_statistics(p[m], data['labels'][i:i + 1], C)
            S_ref[m] += s
        counts_ref += cnt
    S, counts = net._get_sufficient_statistic(data)
    assert np.array_equal(counts, counts_ref)                       # integer work: exact
    assert np.array_equal(params['class_counts'], counts_ref)
    for m in net.modalities:
        # |S| ~ 1e6..1e8 at these sizes; fp32 logs summed in fp64 on both sides, the device log is within 1 ulp
        np.testing.assert_allclose(S[m], S_ref[m], rtol=2e-6, atol=1e-2, err_msg=m)
    # ---- fit: host Newton solve on the device statistics vs on the oracle's -------------------------------------
    ref_params = _oracle_fit(S_ref, counts_ref)
    for m in net.modalities:
        assert np.all(params[m] > 0)
        np.testing.assert_allclose(params[m], ref_params[m], rtol=1e-4, atol=1e-5, err_msg=m)
    # ---- fused prediction with the fitted parameters ------------------------------------------------------------
    fused = net.predict(data)
    score = net.predict(data, output_attr='fused_score')
    assert fused.shape == (2, h, w) and fused.dtype == np.int64
    prior = fu.dirichlet_prior(params['class_counts'], 'data')
    for i in range(len(fused)):
        ref = fu.dirichlet_fusion([fu.renormalise(probs[i][m]) for m in net.modalities],
                                  [params[m].astype(np.float32) for m in net.modalities], prior, CFG['sigma'])[0]
        scale = np.abs(ref).max()
        assert np.abs(score[i] - ref).max() <= 2e-5 * scale + 2e-3, i
        assert np.array_equal(fused[i], np.argmax(score[i], -1))
        top2 = np.sort(ref, -1)[..., -2:]
        clear = (top2[..., 1] - top2[..., 0]) > 4e-5 * scale + 4e-3
        assert clear.mean() > 0.9
        assert np.array_equal(fused[i][clear], np.argmax(ref, -1)[clear])
    # score() over the fused prediction: conservation laws of the confusion matrix
    _, cm = net.score(data)
    assert cm.sum() == (data['labels'] >= 0).sum()
    assert np.array_equal(cm, fu.confusion_matrix(data['labels'], fused, C).astype(np.float64))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, size, port, out, h, w):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=size)
    from modular_semantic_segmentation_amd import parallel
    net = _make()
    params = net.fit(parallel.shard_data(_data(h, w)))           # one image per rank
    np.savez(out % rank, rgb=params['rgb'], depth=params['depth'], class_counts=params['class_counts'])
    dist.destroy_process_group()


@pytest.mark.parametrize('h,w', [(1024, 2048)])
def test_two_rank_dirichlet_fit_equals_single_process(tmp_path, h, w):
    """configs[3] is data parallel: ranks shard the images, `_allreduce_statistics` sums 2*C*C + C numbers, and
    every rank runs the same host fit (dirichlet_mix.py:175-257 has one process see all batches)."""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    out = str(tmp_path / 'fit%d.npz')
    mp.spawn(_worker, args=(2, _free_port(), out, h, w), nprocs=2, join=True)
    r0, r1 = np.load(out % 0), np.load(out % 1)
    single = _make().fit(_data(h, w))
    for key in ('rgb', 'depth', 'class_counts'):
        assert np.array_equal(r0[key], r1[key]), key              # both ranks fitted the same reduced statistics
    assert np.array_equal(r0['class_counts'], single['class_counts'])
    for m in ('rgb', 'depth'):
        # same per-image partial sums, added in a different order (fp64): the fit moves by rounding noise only
        np.testing.assert_allclose(r0[m], single[m], rtol=1e-9, atol=1e-12, err_msg=m)
