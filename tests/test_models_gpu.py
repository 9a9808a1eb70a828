"""End-to-end parity of the model classes (BaseModel API) against the CPU oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import fcn_oracle as fo
from oracle import fusion_oracle as fu

C, U = 12, 64
H, W = 64, 96


def _desc(rgb=True, depth=True):
    dtypes, shapes = {'labels': 'int32'}, {'labels': (None, None)}
    if rgb:
        dtypes['rgb'], shapes['rgb'] = 'float32', (None, None, 3)
    if depth:
        dtypes['depth'], shapes['depth'] = 'float32', (None, None, 1)
    return (dtypes, shapes, C)


def _data(n, seed=0):
    rng = np.random.default_rng(seed)
    return {'rgb': rng.integers(0, 256, (n, H, W, 3)).astype(np.float32),
            'depth': rng.integers(0, 65536, (n, H, W, 1)).astype(np.float32),
            'labels': rng.integers(-1, C, (n, H, W)).astype(np.int32)}


def _weights(tmp_path, prefix, cin, seed, scale_first):
    w = fo.init_fcn_weights(prefix, cin, U, C, seed=seed, bias_scale=0.02)
    # raw 0..65535 depth through Glorot weights would dwarf the biases; scale conv1_1 like a trained net would
    w['%s/conv1_1/kernel' % prefix] *= scale_first
    # a He-like gain keeps activations O(1) through 13 relu layers so that the logits are not degenerate
    for k in w:
        if k.endswith('/kernel') and 'upscore' not in k and 'conv1_1' not in k:
            w[k] *= 1.6
    path = os.path.join(str(tmp_path), prefix + '.npz')
    np.savez(path, **w)
    return w, path


@pytest.fixture(scope='module')
def gpu():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')


def _check_logits_and_labels(got_score, got_label, ref_score, what):
    scale = np.abs(ref_score).max()
    err = np.abs(got_score - ref_score).max() / scale
    # stated tolerance for bf16 storage + fp32 accumulation against the bf16-policy oracle
    from tolerances import LOGIT_TOL_VS_POLICY
    assert err < LOGIT_TOL_VS_POLICY['bf16'], '%s: logits differ by %.3g of max|logit|' % (what, err)
    # labels: bit-exact against the oracle's softmax+argmax fed the SAME logits
    assert np.array_equal(got_label, fo.argmax_last(fo.softmax(got_score))), what
    ref_label = fo.argmax_last(fo.softmax(ref_score))
    agree = (got_label == ref_label).mean()
    top2 = np.sort(ref_score, -1)[..., -2:]
    clear = (top2[..., 1] - top2[..., 0]) > 4e-2 * scale
    assert np.array_equal(got_label[clear], ref_label[clear]), what
    assert agree > 0.97, '%s: label agreement %.4f' % (what, agree)
    return agree


def test_simple_fcn_predict_score_and_weights_io(gpu, tmp_path):
    from modular_semantic_segmentation_amd import get_model
    data = _data(3)
    w, path = _weights(tmp_path, 'rgb', 3, 1, 0.02)
    with get_model('fcn')('rgb', _desc(depth=False), 'rgb', output_dir=str(tmp_path), num_units=U,
                          batch_normalization=False, batchsize=2) as net:
        net.import_weights(path, warnings=False)
        label = net.predict(data)
        score = net.predict(data, output_attr='score')
        prob = net.predict(data, output_attr='prob')
        measures, cm = net.score(data)
        out = net.export_weights()
    assert label.dtype == np.int64 and label.shape == (3, H, W)
    ref = fo.fcn_forward(data['rgb'], w, 'rgb', 'bf16')
    _check_logits_and_labels(score, label, ref['score'], 'SimpleFCN rgb')
    np.testing.assert_allclose(prob, fo.softmax(score), rtol=1e-5, atol=1e-7)
    assert np.array_equal(cm, fu.confusion_matrix(data['labels'], label, C).astype(np.float64))
    assert measures['mean_IoU'] == fu.score_measures(cm)['mean_IoU']
    saved = np.load(out)
    np.testing.assert_array_equal(saved['rgb/conv3_2/kernel'], w['rgb/conv3_2/kernel'])
    # fp32 end-to-end reference (what the TF graph computes): report agreement, loose bound
    ref32 = fo.fcn_forward(data['rgb'], w, 'rgb', 'fp32')
    agree32 = (label == fo.argmax_last(fo.softmax(ref32['score']))).mean()
    assert agree32 > 0.9, agree32


def test_intermediate_layers_match_oracle(gpu, tmp_path):
    from modular_semantic_segmentation_amd.fcn import FcnEngine
    data = _data(1, seed=3)
    w, _ = _weights(tmp_path, 'depth', 1, 2, 2e-4)
    eng = FcnEngine('depth', 1, U, C, w)
    x = torch.from_numpy(data['depth']).cuda()
    out = eng.forward(x, want=('score', 'label'), keep_all=True)
    torch.cuda.synchronize()
    names = ['conv1_1', 'conv1_2', 'pool1', 'conv2_2', 'pool2', 'conv3_3', 'pool3', 'conv4_3', 'pool4', 'conv5_3',
             'score_conv4', 'score_conv5', 'fused']
    ref = fo.fcn_forward(data['depth'], w, 'depth', 'bf16', keep=names + ['score'])
    for name in names:
        got = out['layers'][name].interior().float().cpu().numpy()[..., :ref[name].shape[-1]]
        scale = np.abs(ref[name]).max()
        err = np.abs(got - ref[name]).max() / scale
        assert err < 2e-2, '%s: %.3g' % (name, err)
    _check_logits_and_labels(out['score'].cpu().numpy(), out['label'].cpu().numpy(), ref['score'], 'depth expert')
    # the default inference pass (conv1_1 + conv1_2 + pool1 in ONE kernel where the map tiles in 16x32, no full-resolution
    # map kept) gives the SAME bits as the pass above that materialised every layer with separate kernels
    keep = {k: out['layers'][k].t.clone() for k in ('pool1', 'pool2', 'fused')}
    score_all = out['score'].clone()
    out2 = eng.forward(x, want=('score', 'label'))
    torch.cuda.synchronize()
    assert 'conv1_1' not in out2['layers'] and 'conv1_2' not in out2['layers']          # the fused first pair ran
    for k, v in keep.items():
        assert torch.equal(out2['layers'][k].t, v), k
    assert torch.equal(out2['score'], score_all)


def test_bayes_fusion_model(gpu, tmp_path, golden_dir):
    from modular_semantic_segmentation_amd import get_model
    g = np.load(os.path.join(golden_dir, 'notebook_868.npz'))
    data = _data(2, seed=5)
    wr, pr = _weights(tmp_path, 'rgb', 3, 1, 0.02)
    wd, pd = _weights(tmp_path, 'depth', 1, 2, 2e-4)
    cms = {'rgb': g['cm_rgb'], 'depth': g['cm_depth']}
    net = get_model('bayes_fusion')(data_description=_desc(), confusion_matrices=cms, num_units=U,
                                    prefixes={'rgb': 'rgb', 'depth': 'depth'}, num_channels={'rgb': 3, 'depth': 1},
                                    expert_model='fcn', class_prior='data', batchsize=2)
    net.import_weights(pr, warnings=False)
    net.import_weights(pd, warnings=False)
    fused = net.predict(data)                        # default path: the fused two-expert head kernel
    assert net.expert_outputs is None
    got_score = net.predict(data, output_attr='fused_score')      # unfused path: expert label maps materialised
    la = net.expert_outputs['rgb']['classification'].cpu().numpy()
    lb = net.expert_outputs['depth']['classification'].cpu().numpy()
    mats = [cms['rgb'].astype('float32').T, cms['depth'].astype('float32').T]
    ref_score, _, _ = fu.bayes_fusion([la, lb], mats, 'data')
    np.testing.assert_allclose(got_score, ref_score, rtol=1e-6, atol=1e-5)
    assert np.array_equal(fused, np.argmax(got_score, -1))
    measures, cm = net.score(data)
    assert np.array_equal(cm, fu.confusion_matrix(data['labels'], fused, C).astype(np.float64))
    # the lookup-table variant decides identically wherever the fp32 score is not a near-tie
    net.config['decision_matrix'] = True
    fused_lut = net.predict(data)
    top2 = np.sort(ref_score, -1)[..., -2:]
    clear = (top2[..., 1] - top2[..., 0]) > 1e-4
    assert np.array_equal(fused_lut[clear], fused[clear])
    net.close()


def test_functional_fcn_entry_point(gpu, tmp_path):
    """fcn() as experiments/timing.py calls it: layer dict incl. score / prob / classification."""
    from modular_semantic_segmentation_amd.simple_fcn import fcn
    w, _ = _weights(tmp_path, 'rgb', 3, 1, 0.02)
    data = _data(1, seed=9)
    x = torch.from_numpy(data['rgb']).cuda()
    layers = fcn(x, 'rgb', U, C, variables=w)
    torch.cuda.synchronize()
    assert set(['conv1_1', 'pool4', 'conv5_3', 'fused', 'score', 'prob', 'classification']) <= set(layers)
    ref = fo.fcn_forward(data['rgb'], w, 'rgb', 'bf16')['score']
    _check_logits_and_labels(layers['score'].cpu().numpy(), layers['classification'].cpu().numpy(), ref, 'fcn()')
    again = fcn(x, 'rgb', U, C, variables=w)          # AUTO_REUSE: same engine, same result
    assert torch.equal(again['classification'], layers['classification'])


def test_dirichlet_fusion_fit_and_predict(gpu, tmp_path):
    from modular_semantic_segmentation_amd import get_model
    data = _data(4, seed=6)
    wr, pr = _weights(tmp_path, 'rgb', 3, 1, 0.02)
    wd, pd = _weights(tmp_path, 'depth', 1, 2, 2e-4)
    cfg = dict(data_description=_desc(), modalities=['rgb', 'depth'], num_channels={'rgb': 3, 'depth': 1},
               num_units=U, expert_model='fcn', class_prior='data', sigma=1.0, delta=1e-2, beta=1e-2, batchsize=2)
    net = get_model('dirichlet_fusion')(**cfg)
    net.import_weights(pr, warnings=False)
    net.import_weights(pd, warnings=False)
    with pytest.raises(UserWarning):
        net.predict(data)
    params = net.fit(data)
    assert params['rgb'].shape == (C, C) and params['class_counts'].sum() == (data['labels'] >= 0).sum()
    fused = net.predict(data)
    score = net.predict(data, output_attr='fused_score')
    probs = [net.probs[m].cpu().numpy() for m in ('rgb', 'depth')]       # last batch
    last = slice(2, 4)
    prior = fu.dirichlet_prior(params['class_counts'], 'data')
    ref = fu.dirichlet_fusion([fu.renormalise(p) for p in probs], [params['rgb'], params['depth']], prior, 1.0)
    np.testing.assert_allclose(score[last], ref, rtol=1e-4, atol=2e-2)
    assert np.array_equal(fused, np.argmax(score, -1))
    # sufficient statistics on the GPU == oracle statistics of the same probabilities
    S, counts = net._get_sufficient_statistic(data)
    Sref = np.zeros((C, C))
    for b in (slice(0, 2), slice(2, 4)):
        net.predict({k: v[b] for k, v in data.items()}, output_attr='fused_score')
        p = net.probs['rgb'].cpu().numpy()
        Sref += fu.sufficient_statistics(p, data['labels'][b], C)[0]
    np.testing.assert_allclose(S['rgb'], Sref, rtol=1e-6, atol=1e-3)
    # a model constructed from the fitted parameters predicts the same
    net2 = get_model('dirichlet_mix')(dirichlet_params=params, **cfg)
    net2.import_weights(pr, warnings=False)
    net2.import_weights(pd, warnings=False)
    assert np.array_equal(net2.predict(data), fused)


def test_hip_graph_replay_matches_eager(gpu, tmp_path, golden_dir):
    from modular_semantic_segmentation_amd import get_model
    g = np.load(os.path.join(golden_dir, 'notebook_868.npz'))
    net = get_model('bayes_fusion')(data_description=_desc(), confusion_matrices={'rgb': g['cm_rgb'], 'depth': g['cm_depth']},
                                    num_units=U, prefixes={'rgb': 'rgb', 'depth': 'depth'},
                                    num_channels={'rgb': 3, 'depth': 1}, expert_model='fcn', batchsize=2, seed=4)
    a, b = _data(2, seed=11), _data(2, seed=12)
    eager_a, eager_b = net.predict(a), net.predict(b)
    net.capture_graph({k: torch.from_numpy(v).cuda() for k, v in a.items()})
    assert np.array_equal(net.predict(b), eager_b)
    assert np.array_equal(net.predict(a), eager_a)
    # a different batch size falls back to eager launches
    assert np.array_equal(net.predict({k: v[:1] for k, v in a.items()}), eager_a[:1])


@pytest.mark.parametrize('deconv_shift', [False, True])
def test_simple_fcn_with_batch_normalization_inference(gpu, tmp_path, deconv_shift):
    """batch_normalization=True: moving statistics are folded into (W, b) at load time; parity against the
    oracle's conv -> BN -> relu (custom_layers.py:126-137).  deconv_shift: the batch norms after the two bilinear
    deconvs also carry a shift (and one negative scale), which takes the affine x2 kernel and the general,
    un-commuted decoder head instead of the folded forms."""
    from modular_semantic_segmentation_amd import get_model
    w, _ = _weights(tmp_path, 'rgb', 3, 1, 0.02)
    rng = np.random.default_rng(5)
    for key in list(w):
        if key.endswith('/kernel'):
            layer = key[:-len('/kernel')]
            deconv = 'upscore' in key
            c = w[key].shape[2] if deconv else w[key].shape[3]
            w[layer + '/gamma'] = rng.uniform(0.7, 1.3, c).astype(np.float32)
            w[layer + '/moving_variance'] = rng.uniform(0.6, 1.4, c).astype(np.float32)
            # a scale-only batch norm after a bilinear deconv is folded into the neighbouring 1x1 conv
            plain = deconv and not deconv_shift
            w[layer + '/beta'] = np.zeros(c, np.float32) if plain else (0.05 * rng.standard_normal(c)).astype(np.float32)
            w[layer + '/moving_mean'] = np.zeros(c, np.float32) if plain else \
                (0.05 * rng.standard_normal(c)).astype(np.float32)
            if deconv and deconv_shift:
                w[layer + '/gamma'][3] *= -1.0
    path = os.path.join(str(tmp_path), 'bn.npz')
    np.savez(path, **w)
    data = _data(1, seed=21)
    net = get_model('fcn')('rgb', _desc(depth=False), 'rgb', num_units=U, batch_normalization=True, batchsize=1)
    assert 'rgb/conv3_2/moving_variance' in net.variables
    net.import_weights(path, warnings=False)
    score = net.predict(data, output_attr='score')
    label = net.predict(data)
    ref = fo.fcn_forward(data['rgb'], w, 'rgb', 'bf16')['score']
    _check_logits_and_labels(score, label, ref, 'SimpleFCN with BN')
    # and the fold really matters: ignoring the BN variables gives different logits
    plain = {k: v for k, v in w.items() if k.rsplit('/', 1)[1] in ('kernel', 'bias')}
    ref_plain = fo.fcn_forward(data['rgb'], plain, 'rgb', 'bf16')['score']
    assert np.abs(ref_plain - ref).max() > 0.05 * np.abs(ref).max()


def test_odd_sizes_num_units_20_classes_14(gpu, tmp_path):
    """num_units not a multiple of 64 (experiment 868 uses 20), 14 classes (Synthia), image sides that are
    multiples of 16 but not of the 32-pixel tiles."""
    from modular_semantic_segmentation_amd.fcn import FcnEngine
    u, c, h, w = 20, 14, 48, 80
    wts = fo.init_fcn_weights('depth', 1, u, c, seed=7, bias_scale=0.02)
    wts['depth/conv1_1/kernel'] *= 2e-4
    for k in wts:
        if k.endswith('/kernel') and 'upscore' not in k and 'conv1_1' not in k:
            wts[k] *= 1.6
    rng = np.random.default_rng(8)
    x = rng.integers(0, 65536, (2, h, w, 1)).astype(np.float32)
    eng = FcnEngine('depth', 1, u, c, wts)
    out = eng.forward(torch.from_numpy(x).cuda(), want=('score', 'prob', 'label'))
    torch.cuda.synchronize()
    ref = fo.fcn_forward(x, wts, 'depth', 'bf16')['score']
    assert out['score'].shape == (2, h, w, c)
    _check_logits_and_labels(out['score'].cpu().numpy(), out['label'].cpu().numpy(), ref, 'U=20 C=14 48x80')
    np.testing.assert_allclose(out['prob'].cpu().numpy(), fo.softmax(out['score'].cpu().numpy()), rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize('decoder_bn', ['shifted', 'scale_only'])
def test_fusion_fcn_joint_baseline(gpu, tmp_path, decoder_bn):
    """fusion_fcn() (fusion_fcn.py:11-40): two VGG16 trunks, channel concat, fused 1x1 score convs, decoder with
    its default batch norm -- functional entry point and FusionFCN model against the oracle restatement."""
    from modular_semantic_segmentation_amd import get_model
    from modular_semantic_segmentation_amd.fusion_fcn import fusion_fcn
    prefixes, nch = {'rgb': 'rgb', 'depth': 'depth'}, {'rgb': 3, 'depth': 1}
    w = fo.init_fusion_fcn_weights(prefixes, nch, U, C, seed=3, bias_scale=0.02)
    w['rgb_conv1_1/kernel'] *= 0.02
    w['depth_conv1_1/kernel'] *= 2e-4
    for k in w:
        if k.endswith('/kernel') and 'upscore' not in k and 'conv1_1' not in k:
            w[k] *= 1.6
    if decoder_bn == 'scale_only':
        w['fused/upscore/beta'][:] = 0
        w['fused/upscore/moving_mean'][:] = 0
    data = _data(2, seed=31)
    ref = fo.fusion_fcn_forward(data, w, prefixes, 'bf16')['score']
    # functional form, as experiments/timing.py calls it
    x = {m: torch.from_numpy(data[m]).cuda() for m in prefixes}
    layers = fusion_fcn(x, prefixes, U, C, variables=w)
    torch.cuda.synchronize()
    assert layers['concat_conv4'].c == 1024 and layers['rgb']['conv3_2'].c == 256
    _check_logits_and_labels(layers['score'].cpu().numpy(), layers['classification'].cpu().numpy(), ref,
                             'fusion_fcn %s' % decoder_bn)
    # model class: import the same weights from an npz in the reference's variable-name schema
    path = os.path.join(str(tmp_path), 'fusion.npz')
    np.savez(path, **w)
    with get_model('fusion_fcn')(prefixes, nch, U, C, batchsize=2) as net:
        assert sorted(net.variables) == sorted(w)
        net.import_weights(path, warnings=False)
        label = net.predict(data)
        score = net.predict(data, output_attr='score')
        _check_logits_and_labels(score, label, ref, 'FusionFCN %s' % decoder_bn)
        measures, cm = net.score(data)
        assert cm.sum() == (data['labels'] >= 0).sum()
        out = net.export_weights(str(tmp_path))
        assert np.array_equal(np.load(out)['fused/score/kernel'], w['fused/score/kernel'])


def test_fusion_fcn_with_imported_dense_deconv_kernels(gpu):
    """fusion_fcn() with imported deconv kernels that are not the bilinear constant (the joint model's deconv2d is the same
    general tf.layers.conv2d_transpose, fusion_fcn.py:26-31): the dense transposed-conv path of the joint engine -- x2 deconv +
    add into `features`, x8 deconv -> batch norm -> relu at full resolution, per-pixel score conv -- against the oracle."""
    from modular_semantic_segmentation_amd.fusion_fcn import FusionFcnEngine
    prefixes, nch = {'rgb': 'rgb', 'depth': 'depth'}, {'rgb': 3, 'depth': 1}
    w = fo.init_fusion_fcn_weights(prefixes, nch, U, C, seed=5, bias_scale=0.02)
    w['rgb_conv1_1/kernel'] *= 0.02
    w['depth_conv1_1/kernel'] *= 2e-4
    for k in w:
        if k.endswith('/kernel') and 'upscore' not in k and 'conv1_1' not in k:
            w[k] *= 1.6
    rng = np.random.default_rng(9)
    for name in ('fused_upscore_conv5', 'fused/upscore'):
        kern = w[name + '/kernel']
        w[name + '/kernel'] = (kern + 0.02 * rng.standard_normal(kern.shape) * np.abs(kern).max()).astype(np.float32)
    data = _data(2, seed=33)
    eng = FusionFcnEngine(prefixes, nch, U, C, w)
    assert sorted(eng.dense_deconv) == ['fused/upscore', 'fused_upscore_conv5']
    out = eng.forward({m: torch.from_numpy(data[m]).cuda() for m in prefixes}, want=('score', 'prob', 'label'))
    torch.cuda.synchronize()
    ref = fo.fusion_fcn_forward(data, w, prefixes, 'bf16', keep=['features', 'score'])
    feat = out['layers']['features'].interior().float().cpu().numpy()[..., :U]
    assert np.abs(feat - ref['features']).max() < 2e-2 * np.abs(ref['features']).max()
    _check_logits_and_labels(out['score'].cpu().numpy(), out['label'].cpu().numpy(), ref['score'], 'fusion_fcn dense deconv')
    np.testing.assert_allclose(out['prob'].cpu().numpy().sum(-1), 1.0, atol=1e-5)
    # only the x8 kernel dense (the x2 one bilinear) takes the interpolation kernel for x2 and the dense path for x8
    w2 = dict(w)
    w2['fused_upscore_conv5/kernel'] = fo.init_fusion_fcn_weights(prefixes, nch, U, C, seed=5)['fused_upscore_conv5/kernel']
    eng2 = FusionFcnEngine(prefixes, nch, U, C, w2)
    assert sorted(eng2.dense_deconv) == ['fused/upscore']
    out2 = eng2.forward({m: torch.from_numpy(data[m]).cuda() for m in prefixes}, want=('score', 'label'))
    ref2 = fo.fusion_fcn_forward(data, w2, prefixes, 'bf16')['score']
    _check_logits_and_labels(out2['score'].cpu().numpy(), out2['label'].cpu().numpy(), ref2, 'fusion_fcn dense x8 only')


def test_experiment_flows_without_sacred(gpu, tmp_path):
    """experiments.py: the reference's fit-and-evaluate flows (experiments/bayes_fusion.py:146-195,
    dirichlet_fusion.py:58-81, training.py, evaluation.py) on a synthetic dict-of-arrays dataset."""
    from modular_semantic_segmentation_amd import experiments as ex
    w_rgb, p_rgb = _weights(tmp_path, 'rgb', 3, 1, 0.02)
    w_dep, p_dep = _weights(tmp_path, 'depth', 1, 2, 2e-4)
    testset = _data(6, seed=41)
    measure_set, test_set = ex.split_test_data(testset)
    assert len(measure_set['labels']) == 3 and len(test_set['labels']) == 3
    assert sorted(np.concatenate([measure_set['labels'], test_set['labels']]).sum(axis=(1, 2)).tolist()) == \
        sorted(testset['labels'].sum(axis=(1, 2)).tolist())
    net_config = {'expert_model': 'fcn', 'prefixes': {'rgb': 'rgb', 'depth': 'depth'}, 'num_units': U,
                  'num_channels': {'rgb': 3, 'depth': 1}, 'class_prior': 'data', 'batchsize': 3}
    weights = {'rgb': p_rgb, 'depth': p_dep}
    info = ex.fit_and_evaluate_bayes_fusion(net_config, _desc(), measure_set, test_set, weights)
    n_valid = (test_set['labels'] >= 0).sum()
    assert info['confusion_matrix'].sum() == n_valid
    assert set(info['measurements']) == {'rgb', 'depth', 'fusion'}
    # the experts' confusion matrices on the measurement set are exactly what the oracle counts for their label maps
    for m, w in (('rgb', w_rgb), ('depth', w_dep)):
        assert info['confusion_matrices'][m].sum() == (measure_set['labels'] >= 0).sum()
    # fusing with these matrices: same decisions as the oracle's LUT built from them (where the decision is clear)
    cms = [info['confusion_matrices'][m].astype('float32').T for m in ('rgb', 'depth')]
    lut = fu.bayes_decision_matrix(cms, 'data')
    assert lut.shape == (C, C)
    dcfg = {'expert_model': 'fcn', 'modalities': ['rgb', 'depth'], 'prefixes': {'rgb': 'rgb', 'depth': 'depth'},
            'num_units': U, 'num_channels': {'rgb': 3, 'depth': 1}, 'class_prior': 'data', 'sigma': 1.0, 'delta': 1e-2,
            'beta': 1e-2, 'batchsize': 3}
    dinfo = ex.fit_and_evaluate_dirichlet_fusion(dcfg, _desc(), measure_set, test_set, [p_rgb, p_dep])
    assert dinfo['confusion_matrix'].sum() == n_valid
    assert dinfo['dirichlet_params']['rgb'].shape == (C, C) and np.all(dinfo['dirichlet_params']['rgb'] > 0)
    # training flow: a few iterations, export, evaluate
    tinfo = ex.train_and_evaluate('fcn', {'prefix': 'rgb', 'modality': 'rgb', 'num_units': U, 'batch_normalization': False,
                                          'batchsize': 2, 'learning_rate': 1e-4},
                                  _desc(depth=False), {k: measure_set[k] for k in ('rgb', 'labels')},
                                  {k: test_set[k] for k in ('rgb', 'labels')}, 3, starting_weights=p_rgb,
                                  output_dir=str(tmp_path))
    assert os.path.exists(tinfo['weights']) and tinfo['confusion_matrix'].sum() == n_valid


def test_fusion_fcn_and_bn_training_with_padded_units(gpu, tmp_path):
    """num_units = 20 (padded to 64 lanes inside the engines) and 14 classes: the joint model against the oracle,
    and a few batch-norm training steps (the padding channels must stay exactly zero and the loss must fall)."""
    from modular_semantic_segmentation_amd import get_model
    from modular_semantic_segmentation_amd.fusion_fcn import FusionFcnEngine
    u, c, h, w = 20, 14, 48, 80
    prefixes, nch = {'rgb': 'rgb', 'depth': 'depth'}, {'rgb': 3, 'depth': 1}
    wts = fo.init_fusion_fcn_weights(prefixes, nch, u, c, seed=9, bias_scale=0.02)
    wts['rgb_conv1_1/kernel'] *= 0.02
    wts['depth_conv1_1/kernel'] *= 2e-4
    for k in wts:
        if k.endswith('/kernel') and 'upscore' not in k and 'conv1_1' not in k:
            wts[k] *= 1.6
    rng = np.random.default_rng(10)
    data = {'rgb': rng.integers(0, 256, (2, h, w, 3)).astype(np.float32),
            'depth': rng.integers(0, 65536, (2, h, w, 1)).astype(np.float32)}
    eng = FusionFcnEngine(prefixes, nch, u, c, wts)
    out = eng.forward({m: torch.from_numpy(v).cuda() for m, v in data.items()}, want=('score', 'label'))
    torch.cuda.synchronize()
    ref = fo.fusion_fcn_forward(data, wts, prefixes, 'bf16')['score']
    got = out['score'].cpu().numpy()
    assert got.shape == (2, h, w, c)
    assert np.abs(got - ref).max() < 2e-2 * np.abs(ref).max()
    # batch-norm training with padded units
    desc = ({'rgb': 'float32', 'labels': 'int32'}, {'rgb': (None, None, 3), 'labels': (None, None)}, c)
    net = get_model('fcn')('rgb', desc, 'rgb', num_units=u, batch_normalization=True, batchsize=2, learning_rate=1e-3,
                           trainer='adam', seed=3)
    batch = {'rgb': data['rgb'], 'labels': rng.integers(-1, c, (2, h, w)).astype(np.int32)}
    first = net._train_batch(batch)
    for _ in range(8):
        last = net._train_batch(batch)
    assert np.isfinite(last) and last < first
    tr = net.trainer
    assert float(tr.view(tr.param, 'score_conv4', 'kernel')[..., u:].abs().max()) == 0.0
    assert net.predict(batch).shape == (2, h, w)


def test_bn_training_with_units_between_129_and_192(gpu):
    """num_units = 160 used to be padded to 192 lanes, a width no batch-norm kernel takes (batchnorm.hip: c >= 64,
    2048 % c == 0): batch-norm training raised XV_ESHAPE.  fcn.padded_units now pads to 64 / 128 / 256 lanes;
    the step must train and keep the padding lanes at zero, the inference engine must accept the trained variables."""
    from modular_semantic_segmentation_amd import get_model, trainer
    from modular_semantic_segmentation_amd.fcn import padded_units
    assert not trainer._ups8_channels_ok(192) and trainer._ups8_channels_ok(64) and trainer._ups8_channels_ok(128)
    assert [padded_units(u) for u in (1, 64, 65, 128, 129, 160, 192, 256, 257, 300)] == [64, 64, 128, 128, 256, 256, 256, 256, 320, 320]
    u, c, h, w = 160, 12, 32, 48
    rng = np.random.default_rng(21)
    desc = ({'rgb': 'float32', 'labels': 'int32'}, {'rgb': (None, None, 3), 'labels': (None, None)}, c)
    net = get_model('fcn')('rgb', desc, 'rgb', num_units=u, batch_normalization=True, batchsize=2, learning_rate=1e-3,
                           trainer='adam', seed=5)
    batch = {'rgb': rng.integers(0, 256, (2, h, w, 3)).astype(np.float32),
             'labels': rng.integers(-1, c, (2, h, w)).astype(np.int32)}
    first = net._train_batch(batch)
    for _ in range(6):
        last = net._train_batch(batch)
    assert net.engine.Up == 256
    assert np.isfinite(last) and last < first
    tr = net.trainer
    assert float(tr.view(tr.param, 'score_conv4', 'kernel')[..., u:].abs().max()) == 0.0
    assert net.predict(batch).shape == (2, h, w)


def test_dataset_readers_drive_training_and_fusion_flows(gpu, tmp_path):
    """the SynthiaCityscapes reader's streams (per-image dicts, the reference's tf.data
    role) feed fit / score and the Bayes-fusion experiment flow; a stream and the stacked numpy batch of the
    same items give the same confusion matrix."""
    import dataset_fixtures as fx
    from modular_semantic_segmentation_amd import experiments as ex, get_model
    from modular_semantic_segmentation_amd.datasets import get_dataset
    fx.build_synthia_tree(str(tmp_path / 'synthia'), image_hw=(70, 100))
    Data = get_dataset('synthia_cityscapes')
    data = Data(base_path=str(tmp_path / 'synthia'),
                augmentation={'crop': [1, 64], 'scale': False, 'vflip': .3, 'hflip': False, 'gamma': [.4, .3, 1.2],
                              'rotate': False, 'shear': False, 'contrast': [.3, .5, 1.5], 'brightness': [.2, -40, 40]})
    desc = Data.get_data_description()
    assert desc[2] == 12
    w_rgb, p_rgb = _weights(tmp_path, 'rgb', 3, 1, 0.02)
    w_dep, p_dep = _weights(tmp_path, 'depth', 1, 2, 2e-4)
    with get_model('fcn')('rgb', desc, 'rgb', num_units=U, batch_normalization=False, batchsize=2,
                          learning_rate=1e-5) as net:
        net.import_weights(p_rgb)
        net.fit(data.get_trainset(), 4, output=False, validation_dataset=data.get_validation_set(num_items=2))
        assert net.global_step == 4
        m_stream, cm_stream = net.score(data.get_measureset())
        m_batch, cm_batch = net.score(data.get_measureset(tf_dataset=False))
        assert np.array_equal(cm_stream, cm_batch) and cm_stream.sum() == len(data.measureset) * 64 * 96
        pred = net.predict(data.get_testset(num_items=3))
        assert pred.shape == (3, 64, 96) and pred.dtype == np.int64
    net_config = {'expert_model': 'fcn', 'prefixes': {'rgb': 'rgb', 'depth': 'depth'}, 'num_units': U,
                  'num_channels': {'rgb': 3, 'depth': 1}, 'class_prior': 'data', 'batchsize': 2}
    info = ex.fit_and_evaluate_bayes_fusion(net_config, desc, data.get_measureset(), data.get_testset(),
                                            {'rgb': p_rgb, 'depth': p_dep})
    assert info['confusion_matrix'].sum() == len(data.testset) * 64 * 96
    assert info['confusion_matrices']['rgb'].sum() == len(data.measureset) * 64 * 96


def test_fcn_with_imported_dense_deconv_kernels(gpu):
    """An imported deconv kernel that is not the bilinear constant (custom_layers.deconv2d is a general
    tf.layers.conv2d_transpose) takes the dense transposed-conv path: x2 deconv + add into `fused`, x8 deconv + relu at
    full resolution, per-pixel score conv, softmax, argmax -- against the oracle's dense deconv."""
    from modular_semantic_segmentation_amd.fcn import FcnEngine
    w = fo.init_fcn_weights('rgb', 3, U, C, seed=3, bias_scale=0.02)
    w['rgb/conv1_1/kernel'] *= 0.02
    for k in w:
        if k.endswith('/kernel') and 'upscore' not in k and 'conv1_1' not in k:
            w[k] *= 1.6
    rng = np.random.default_rng(5)
    for name in ('upscore_conv5', 'upscore'):
        kern = w['rgb/%s/kernel' % name]
        w['rgb/%s/kernel' % name] = (kern + 0.02 * rng.standard_normal(kern.shape) * np.abs(kern).max()).astype(np.float32)
    x = rng.integers(0, 256, (2, 64, 96, 3)).astype(np.float32)
    eng = FcnEngine('rgb', 3, U, C, w)
    assert sorted(eng.dense_deconv) == ['upscore', 'upscore_conv5']
    out = eng.forward(torch.from_numpy(x).cuda(), want=('score', 'prob', 'label'))
    torch.cuda.synchronize()
    ref = fo.fcn_forward(x, w, 'rgb', 'bf16', keep=['fused', 'upscore', 'score'])
    fused = out['layers']['fused'].interior().float().cpu().numpy()[..., :U]
    assert np.abs(fused - ref['fused']).max() < 2e-2 * np.abs(ref['fused']).max()
    up = out['layers']['upscore'].interior().float().cpu().numpy()[..., :U]
    assert np.abs(up - ref['upscore']).max() < 2e-2 * np.abs(ref['upscore']).max()
    _check_logits_and_labels(out['score'].cpu().numpy(), out['label'].cpu().numpy(), ref['score'], 'dense deconv')
    np.testing.assert_allclose(out['prob'].cpu().numpy().sum(-1), 1.0, atol=1e-5)


@pytest.mark.parametrize('kind', ['bayes', 'dirichlet'])
def test_fused_head_equals_unfused_path(gpu, tmp_path, golden_dir, kind):
    """The default prediction of a two-expert fusion model runs ONE fused head kernel after the trunks (logits, per-expert
    softmax / argmax and the Bayes / Dirichlet mix in registers).  Its labels must equal, bit for bit, those of the
    unfused path (decoder head per expert -> label / probability maps in HBM -> fusion kernel), which is the path the
    oracle comparisons of this file pin; serial and two-stream execution agree as well."""
    from modular_semantic_segmentation_amd import get_model
    g = np.load(os.path.join(golden_dir, 'notebook_868.npz'))
    data = _data(3, seed=21)
    _, pr = _weights(tmp_path, 'rgb', 3, 1, 0.02)
    _, pd = _weights(tmp_path, 'depth', 1, 2, 2e-4)
    common = dict(data_description=_desc(), num_units=U, num_channels={'rgb': 3, 'depth': 1}, expert_model='fcn',
                  class_prior='data', batchsize=3)
    rng = np.random.default_rng(4)

    def make(**kw):
        if kind == 'bayes':
            net = get_model('bayes_fusion')(confusion_matrices={'rgb': g['cm_rgb'], 'depth': g['cm_depth']},
                                            prefixes={'rgb': 'rgb', 'depth': 'depth'}, **common, **kw)
        else:
            params = {'rgb': rng.uniform(0.5, 4.0, (C, C)), 'depth': rng.uniform(0.5, 4.0, (C, C)),
                      'class_counts': g['cm_depth'].sum(1)}
            net = get_model('dirichlet_fusion')(dirichlet_params=params, modalities=['rgb', 'depth'], sigma=1.0, delta=1e-2,
                                                beta=1e-2, **common, **kw)
            make.params = params
        net.import_weights(pr, warnings=False)
        net.import_weights(pd, warnings=False)
        return net

    fused_net = make()
    got = fused_net.predict(data)
    plain = fused_net
    plain.config['fused_head'] = False
    ref = plain.predict(data)
    assert np.array_equal(got, ref)
    plain.config['fused_head'] = True
    plain.concurrent_experts = False
    assert np.array_equal(plain.predict(data), ref)


@pytest.mark.parametrize('c', [3, 5, 12, 14, 16])
def test_fused_head_op_equals_decoder_heads_and_fusion_kernels(gpu, c):
    """Op level, for class counts that fill their template (12, 16) and that do not (3, 5, 14): both experts' 1/8-resolution
    features -> xv_score_lowres -> xv_fused_head_fwd against decoder head per expert (labels / probabilities in HBM) ->
    bayes_fuse / dirichlet_fuse.  The Bayes head decides through a [C][C] table its workgroups build from the log-likelihood
    tables (the sums of bayes_fuse_kernel in its order): labels must be equal bit for bit."""
    from modular_semantic_segmentation_amd import ops
    n, hi, wi, u = 2, 5, 7, 64
    g = torch.Generator().manual_seed(7 * c)
    dev = 'cuda:0'
    feats = [ops.Act.from_dense((torch.rand((n, hi, wi, u), generator=g) * 2).to(dev)) for _ in range(2)]
    ws = [(torch.randn((u, c), generator=g) * 0.3).to(dev) for _ in range(2)]
    bs = [torch.randn(c, generator=g).to(dev) for _ in range(2)]
    cp = (c + 3) // 4 * 4
    S = [torch.zeros((n, hi + 2, wi + 2, cp), device=dev) for _ in range(2)]
    for e in range(2):
        ops.score_lowres(feats[e], ws[e], c, S[e])
    heads = [ops.decoder_head_fwd(feats[e], ws[e], bs[e], c, want_prob=True, want_label=True) for e in range(2)]
    for e in range(2):      # the label alone runs four pixels per thread (decoder_head_label4_kernel): the same labels
        assert torch.equal(ops.decoder_head_fwd(feats[e], ws[e], bs[e], c)['label'], heads[e]['label'])
    loglik = torch.randn((2, c, c), generator=g).to(dev)
    logprior = torch.randn(c, generator=g).to(dev)
    ref, _ = ops.bayes_fuse([heads[0]['label'], heads[1]['label']], loglik, logprior)
    got = ops.fused_head(S[0], S[1], bs[0], bs[1], n, hi, wi, c, loglik, logprior)
    assert ref.unique().numel() > 1
    assert torch.equal(got, ref)
    am1 = (torch.rand((2, c, c), generator=g) - 0.5 + 4 * torch.eye(c)).to(dev)
    lognorm = (torch.randn((2, c), generator=g) * 0.1).to(dev)
    ref, _ = ops.dirichlet_fuse([heads[0]['prob'], heads[1]['prob']], am1, lognorm, logprior)
    got = ops.fused_head(S[0], S[1], bs[0], bs[1], n, hi, wi, c, am1, logprior, lognorm=lognorm)
    assert torch.equal(got, ref)


@pytest.mark.parametrize('c', [4, 8, 12, 16])
@pytest.mark.parametrize('shape', [(1, 3, 5), (2, 6, 4), (3, 7, 9)])
def test_dirichlet_head_packed_form_equals_scalar_form(gpu, c, shape):
    """The fused Dirichlet head for a class count that fills its template (C == CM) runs on packed fp32, four output pixels
    per thread (fused_dirichlet_head_pk_kernel); XV_DIRICHLET_HEAD_PK=0 keeps the scalar one-pixel form, which the test
    above pins to the unfused path.  Same IEEE operations, same summation order: the labels must be equal."""
    from modular_semantic_segmentation_amd import ops
    n, hi, wi = shape
    g = torch.Generator().manual_seed(100 * c + hi)
    S = [torch.zeros((n, hi + 2, wi + 2, c)) for _ in range(2)]
    for t in S:
        t[:, 1:-1, 1:-1] = torch.randn((n, hi, wi, c), generator=g) * 3
    Sa, Sb = (t.to('cuda:0') for t in S)
    ba, bb = torch.randn(c, generator=g).to('cuda:0'), torch.randn(c, generator=g).to('cuda:0')
    am1 = (torch.rand((2, c, c), generator=g) - 0.5 + 4 * torch.eye(c)).to('cuda:0')   # each class likeliest under its own row
    lognorm = (torch.randn((2, c), generator=g) * 0.1).to('cuda:0')
    logprior = (torch.randn(c, generator=g) * 0.1).to('cuda:0')
    old = os.environ.get('XV_DIRICHLET_HEAD_PK')
    try:
        os.environ['XV_DIRICHLET_HEAD_PK'] = '0'
        ref = ops.fused_head(Sa, Sb, ba, bb, n, hi, wi, c, am1, logprior, lognorm=lognorm)
        os.environ['XV_DIRICHLET_HEAD_PK'] = '1'
        got = ops.fused_head(Sa, Sb, ba, bb, n, hi, wi, c, am1, logprior, lognorm=lognorm)
    finally:
        if old is None:
            os.environ.pop('XV_DIRICHLET_HEAD_PK', None)
        else:
            os.environ['XV_DIRICHLET_HEAD_PK'] = old
    assert ref.unique().numel() > 1
    assert torch.equal(got, ref)


def test_paired_expert_launches_equal_per_expert_launches(gpu, tmp_path, golden_dir):
    """From conv4_1 on the two experts of a fusion model share ONE launch per layer (fcn.encoder_layers_pair ->
    xv_conv2d_fwd_pair: whole rounds of workgroups on the persistent conv kernels).  At 768x384 -- conv4 maps tile in 16x32,
    conv5 maps in 24x16 -- the fused labels must equal, bit for bit, those of per-expert launches, from any starting layer,
    with and without the fused head, on one stream and on two."""
    from modular_semantic_segmentation_amd import fcn, get_model, ops
    g = np.load(os.path.join(golden_dir, 'notebook_868.npz'))
    rng = np.random.default_rng(8)
    n, h, w = 4, 384, 768          # 4 x 3 x 3 x 8 conv4 tiles per expert: more than one round of the 256 CUs
    data = {'rgb': rng.integers(0, 256, (n, h, w, 3)).astype(np.float32),
            'depth': rng.integers(0, 65536, (n, h, w, 1)).astype(np.float32)}
    _, pr = _weights(tmp_path, 'rgb', 3, 1, 0.02)
    _, pd = _weights(tmp_path, 'depth', 1, 2, 2e-4)
    net = get_model('bayes_fusion')(data_description=_desc(), confusion_matrices={'rgb': g['cm_rgb'], 'depth': g['cm_depth']},
                                    num_units=U, prefixes={'rgb': 'rgb', 'depth': 'depth'}, num_channels={'rgb': 3, 'depth': 1},
                                    expert_model='fcn', class_prior='data', batchsize=n)
    net.import_weights(pr, warnings=False)
    net.import_weights(pd, warnings=False)
    calls = []
    real = ops.conv2d_fwd_pair

    def counting(*a, **kw):
        ok = real(*a, **kw)
        calls.append(ok)
        return ok
    saved = fcn.GROUP_FROM
    ops.conv2d_fwd_pair = counting
    try:
        fcn.GROUP_FROM = '0'
        ref = net.predict(data)
        assert not calls
        for start, launches in (('conv4_1', 6), ('conv2_1', 11)):
            fcn.GROUP_FROM = start
            del calls[:]
            net._graph = None
            assert np.array_equal(net.predict(data), ref), start
            assert calls == [True] * launches, (start, calls)
        fcn.GROUP_FROM = 'conv4_1'
        net.concurrent_experts = False
        assert np.array_equal(net.predict(data), ref)
        net.concurrent_experts = True
        net.config['fused_head'] = False
        assert np.array_equal(net.predict(data), ref)
        assert len(np.unique(ref)) > 1
        # one image: fewer tiles than CUs, the experts' kernels run side by side on their streams -- no paired section
        del calls[:]
        net._graph = None
        one = {k: v[:1] for k, v in data.items()}
        assert np.array_equal(net.predict(one), ref[:1]) and not calls
    finally:
        fcn.GROUP_FROM = saved
        ops.conv2d_fwd_pair = real
    net.close()


def test_config1_simple_fcn_rgb_256x512_14_classes(gpu, tmp_path):
    """BASELINE.json configs[0]: SimpleFCN RGB-only, Synthia 256x512 (tensor [1,256,512,3]), batch 1, 14 classes -- the
    reference's own CPU-runnable case: the oracle IS that CPU path restated; the HIP path must reproduce its label map
    (wherever the fp32 margin is clear) and its `score` measures on the same weights through the model API."""
    from modular_semantic_segmentation_amd import get_model
    C14 = 14
    rng = np.random.default_rng(14)
    data = {'rgb': rng.integers(0, 256, (1, 256, 512, 3)).astype(np.float32),
            'labels': rng.integers(-1, C14, (1, 256, 512)).astype(np.int32)}
    w = fo.init_fcn_weights('rgb', 3, U, C14, seed=7, bias_scale=0.02)
    w['rgb/conv1_1/kernel'] *= 0.02
    for k in w:
        if k.endswith('/kernel') and 'upscore' not in k and 'conv1_1' not in k:
            w[k] *= 1.6
    desc = ({'rgb': 'float32', 'labels': 'int32'}, {'rgb': (256, 512, 3), 'labels': (256, 512)}, C14)
    net = get_model('fcn')('rgb', desc, 'rgb', num_units=U, batch_normalization=False, batchsize=1)
    net.variables.update(w)
    net._variables_changed()
    pred = net.predict(data)
    score = net.predict(data, output_attr='score')
    assert pred.shape == (1, 256, 512) and pred.dtype == np.int64 and score.shape == (1, 256, 512, C14)
    ref = fo.fcn_forward(data['rgb'], w, 'rgb', 'fp32', keep=['score'])['score']          # the reference's fp32 graph
    scale = np.abs(ref).max()
    assert np.abs(score - ref).max() / scale < 3e-2
    ref_lab = fo.argmax_last(fo.softmax(ref))
    top2 = np.sort(ref, -1)[..., -2:]
    clear = (top2[..., 1] - top2[..., 0]) > 6e-2 * scale
    assert np.array_equal(pred[clear], ref_lab[clear])
    measures, cm = net.score(data)
    assert np.array_equal(cm, fu.confusion_matrix(data['labels'], pred, C14).astype(np.float64))
    ref_measures = fu.score_measures(fu.confusion_matrix(data['labels'], pred, C14))
    assert measures['mean_IoU'] == ref_measures['mean_IoU'] or (np.isnan(measures['mean_IoU']) and np.isnan(ref_measures['mean_IoU']))


def test_mc_dropout_sites_of_encoder_and_decoder(gpu):
    """The dropout branches of encoder() / decoder() (simple_fcn.py:50-62,71-78,124-126: tf.layers.dropout with
    training=True, used by the uncertainty models): kept elements are x / (1 - rate), the rest zero, about `rate` of
    them; 'pool3' also drops after pool4 and 'pool4' alone drops nothing (the reference's key quirk); every call draws
    new masks, a seed reproduces them; without dropout arguments the graph is the plain one."""
    from modular_semantic_segmentation_amd.simple_fcn import encoder, fcn
    w = fo.init_fcn_weights('rgb', 3, U, C, seed=9, bias_scale=0.02)
    w['rgb/conv1_1/kernel'] *= 0.02
    for k in w:
        if k.endswith('/kernel') and 'upscore' not in k and 'conv1_1' not in k:
            w[k] *= 1.6
    x = torch.from_numpy(np.random.default_rng(3).integers(0, 256, (1, 64, 96, 3)).astype(np.float32)).cuda()
    plain = {k: v.interior().float().clone() for k, v in encoder(x, 'rgb', U, variables=w, num_classes=C).items()}
    L = encoder(x, 'rgb', U, 0.5, variables=w, num_classes=C, dropout_layers=['pool3', 'conv5_3'], dropout_seed=7)
    torch.cuda.synchronize()
    assert 'pool3_drop' in L and 'pool4_drop' in L
    p3, d3 = L['pool3'].interior().float(), L['pool3_drop'].interior().float()
    assert torch.equal(p3, plain['pool3'])                        # everything before the first site is untouched
    kept = d3 != 0
    frac = 1.0 - kept.float().mean().item() / max((p3 != 0).float().mean().item(), 1e-9)
    assert 0.45 < frac < 0.55, frac
    assert torch.allclose(d3[kept], (p3[kept] * 2).bfloat16().float())
    assert not L['pool3_drop'].t[:, 0].any() and not L['pool3_drop'].t[:, :, -1].any()        # zero border kept
    assert not torch.equal(L['fused'].interior().float(), plain['fused'])
    d3_first = d3.clone()
    L2 = encoder(x, 'rgb', U, 0.5, variables=w, num_classes=C, dropout_layers=['pool3', 'conv5_3'], dropout_seed=7)
    assert not torch.equal(L2['pool3_drop'].interior().float(), d3_first)                      # a new pass, new masks
    only4 = encoder(x, 'rgb', U, 0.5, variables=w, num_classes=C, dropout_layers=['pool4'])
    assert 'pool4_drop' not in only4 and torch.equal(only4['fused'].interior().float(), plain['fused'])
    out = fcn(x, 'rgb', U, C, variables=w, dropout_rate=0.3, dropout_layers=['features'])
    assert 'features_drop' in out and out['score'].shape == (1, 64, 96, C)
    # the reference's layer dict keeps 'fused' undropped: the decoder drops its input internally (simple_fcn.py:124-126)
    assert torch.equal(out['fused'].interior().float(), plain['fused'])
    assert not torch.equal(out['features_drop'].interior().float(), plain['fused'])
    again = encoder(x, 'rgb', U, variables=w, num_classes=C)                                   # dropout is off again
    assert torch.equal(again['fused'].interior().float(), plain['fused'])
