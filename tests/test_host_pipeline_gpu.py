"""The pipelined host boundary (modular_semantic_segmentation_amd/host_pipeline.py) of predict / score / fit -- the API the
reference's callers use with HOST arrays (xview/models/base_model.py:180-331; its tf.data prefetch: :203-206): pinned
staging by worker threads, uploads and label downloads on copy streams, the step captured into a hipGraph after two
same-shape batches.  Same results as the serial path (XV_HOST_PIPELINE=0), bit for bit: labels, confusion matrices, and the
weights after training steps."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

C, U = 12, 64
H, W = 64, 96


def _desc():
    return ({'rgb': 'float32', 'depth': 'float32', 'labels': 'int32'},
            {'rgb': (None, None, 3), 'depth': (None, None, 1), 'labels': (None, None)}, C)


def _data(n, seed=0):
    rng = np.random.default_rng(seed)
    return {'rgb': rng.integers(0, 256, (n, H, W, 3)).astype(np.float32),
            'depth': rng.integers(0, 65536, (n, H, W, 1)).astype(np.float32),
            'labels': rng.integers(-1, C, (n, H, W)).astype(np.int32)}


@pytest.fixture(scope='module')
def net(golden_dir):
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from modular_semantic_segmentation_amd import get_model
    g = np.load(os.path.join(golden_dir, 'notebook_868.npz'))
    net = get_model('bayes_fusion')(data_description=_desc(), confusion_matrices={'rgb': g['cm_rgb'], 'depth': g['cm_depth']},
                                    num_units=U, prefixes={'rgb': 'rgb', 'depth': 'depth'},
                                    num_channels={'rgb': 3, 'depth': 1}, expert_model='fcn', batchsize=4, seed=4)
    net.variables['depth/conv1_1/kernel'] = net.variables['depth/conv1_1/kernel'] / 256.0
    net._variables_changed()
    return net


def _serial(fn):
    from modular_semantic_segmentation_amd import host_pipeline
    host_pipeline.ENABLED = False
    try:
        return fn()
    finally:
        host_pipeline.ENABLED = True


def test_pipelined_predict_equals_serial_predict(net):
    data = _data(26, seed=3)                       # six full batches of 4 and one of 2
    inputs = {k: v for k, v in data.items() if k != 'labels'}
    net._graph = None
    want = _serial(lambda: net.predict(inputs))
    assert net._graph is None                                       # the serial path captures nothing
    got = net.predict(inputs)
    assert got.dtype == np.int64 and got.shape == (26, H, W) and np.array_equal(got, want)
    assert net._graph is not None                                   # two eager same-shape batches, then the captured step
    again = net.predict(inputs)                                     # replays from the first batch on
    assert np.array_equal(again, want)
    # uint8 / uint16 source arrays are converted while they are staged (one pass), like np.asarray(..., float32) would
    raw = {'rgb': inputs['rgb'].astype(np.uint8), 'depth': inputs['depth'].astype(np.uint16)}
    assert np.array_equal(net.predict(raw), want)
    # any named output travels the same way
    score = net.predict(inputs, output_attr='fused_score')
    assert score.shape == (26, H, W, C) and np.array_equal(np.argmax(score, -1), want)
    # an iterable of per-sample dicts (the tf.data.Dataset case): the total is unknown in advance
    samples = [{k: v[i] for k, v in inputs.items()} for i in range(26)]
    assert np.array_equal(net.predict(iter(samples)), want)
    # inputs already resident in HBM are passed through
    dev = {k: torch.from_numpy(v).cuda() for k, v in inputs.items()}
    assert np.array_equal(net.predict(dev), want)


def test_pipelined_score_equals_serial_score(net):
    data = _data(22, seed=8)
    net._graph = None
    m0, cm0 = _serial(lambda: net.score(data))
    m1, cm1 = net.score(data)
    assert np.array_equal(cm0, cm1) and cm1.sum() == (data['labels'] >= 0).sum()
    assert m0['mean_IoU'] == m1['mean_IoU']
    _, cm2 = net.score(data, max_iterations=3)
    _, cm3 = _serial(lambda: net.score(data, max_iterations=3))
    assert np.array_equal(cm2, cm3) and cm2.sum() == (data['labels'][:12] >= 0).sum()


def test_a_graph_captured_under_other_switches_is_not_replayed(net):
    data = _data(16, seed=5)
    inputs = {k: v for k, v in data.items() if k != 'labels'}
    net._graph = None
    want = net.predict(inputs)
    assert net._graph is not None
    captured = net._graph
    net.config['fused_head'] = False
    try:
        assert np.array_equal(net._predict_batch({k: torch.from_numpy(v[:4]).cuda() for k, v in inputs.items()}).cpu().numpy(),
                              want[:4])
        assert net.expert_outputs is not None          # the unfused path ran (eagerly): per-expert outputs materialised
    finally:
        net.config['fused_head'] = True
    assert net._graph is captured
    net._variables_changed()                            # new weights drop the graph (it replays the old pointers)
    assert net._graph is None


def test_pipelined_fit_equals_serial_fit():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from modular_semantic_segmentation_amd import get_model
    desc = ({'rgb': 'float32', 'labels': 'int32'}, {'rgb': (None, None, 3), 'labels': (None, None)}, C)
    data = {k: v for k, v in _data(6, seed=2).items() if k != 'depth'}
    out = []
    for serial in (True, False):
        net = get_model('fcn')('rgb', desc, 'rgb', num_units=U, batch_normalization=False, batchsize=4, learning_rate=1e-4,
                               trainer='adam', seed=7)
        run = lambda: net.fit(data, 5, output=False)       # noqa: E731  (5 steps: batches wrap around the 6 samples)
        _serial(run) if serial else run()
        net._sync_variables()
        out.append({k: np.array(v) for k, v in net.variables.items()})
    for k in out[0]:
        assert np.array_equal(out[0][k], out[1][k]), k


def test_narrow_labels_kernel():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    import ctypes
    from modular_semantic_segmentation_amd import _lib
    rng = np.random.default_rng(0)
    for n in (1, 7, 8, 9, 4096 + 5, 2 * 384 * 768):
        lab = torch.from_numpy(rng.integers(0, 256, n).astype(np.int64)).cuda()
        out = torch.full((n + 8,), 77, dtype=torch.uint8, device='cuda')
        assert _lib.lib().xv_narrow_labels(ctypes.c_void_p(lab.data_ptr()), n, ctypes.c_void_p(out.data_ptr()), None) == 0
        torch.cuda.synchronize()
        assert torch.equal(out[:n].long(), lab) and bool((out[n:] == 77).all())
