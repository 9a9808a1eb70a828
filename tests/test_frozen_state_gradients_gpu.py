"""Tight training-step parity: every backward op of a training step against the oracle on the step's OWN stored forward
state.  The whole-step comparisons of tests/test_backward_gpu.py run the oracle from the image, so bf16 rounding noise
compounds through 16 layers and the tolerances there are 5-70 %; here each layer's filter gradient, bias gradient and
data gradient is recomputed (torch-CPU fp32, the arithmetic of tf.gradients: Conv2DBackpropFilter, BiasAddGrad,
Conv2DBackpropInput, ReluGrad, MaxPoolGrad, AddN; base_model.py:153-162 over simple_fcn.py:200-214) from the exact bf16
activations and gradient maps the MI355X step left in HBM -- relu masks and pool routes are the GPU's own, so nothing
compounds and a wrong sign / tap / channel in a rarely-hit path has nowhere to hide:

    filter and bias gradients   <= 1e-3 of the tensor's largest entry (fp32 sums in another order)
    data gradients              <= 1 %  of the map's largest entry (one bf16 rounding of the stored map: 2^-9 relative)
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import fcn_oracle as fo

C, U, H, W = 12, 64, 32, 48


@pytest.fixture(scope='module')
def ops():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from modular_semantic_segmentation_amd import ops as _ops
    return _ops


def _nchw(act):
    """Act (bf16 padded NHWC on the device) -> float32 NCHW on the host, exact."""
    return act.interior().float().cpu().permute(0, 3, 1, 2).contiguous()


def _w_oihw(hwio):
    return torch.from_numpy(np.ascontiguousarray(hwio)).permute(3, 2, 0, 1).contiguous()


def _close(got, ref, tol, what):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    scale = np.abs(ref).max() + 1e-30
    err = np.abs(got - ref).max() / scale
    assert err <= tol, '%s: max error %.3g of the largest entry (tolerance %.3g)' % (what, err, tol)
    return err


def _setup(bn, tmp_path):
    from modular_semantic_segmentation_amd import get_model
    rng = np.random.default_rng(0)
    data = {'rgb': rng.integers(0, 256, (2, H, W, 3)).astype(np.float32),
            'labels': rng.integers(-1, C, (2, H, W)).astype(np.int32)}
    w = fo.init_fcn_weights('rgb', 3, U, C, seed=1, bias_scale=0.02)
    w['rgb/conv1_1/kernel'] *= 0.02
    for k in w:
        if k.endswith('/kernel') and 'upscore' not in k and 'conv1_1' not in k:
            w[k] *= 1.6
    desc = ({'rgb': 'float32', 'labels': 'int32'}, {'rgb': (None, None, 3), 'labels': (None, None)}, C)
    net = get_model('fcn')('rgb', desc, 'rgb', num_units=U, batch_normalization=bn, batchsize=2, learning_rate=1e-3,
                           trainer='adam', seed=1)
    net.variables.update(w)
    net._variables_changed()
    tr = net._ensure_trainer()
    tr.keep_all = True      # (batch-norm trainer: keep the activations of the pooled layers for the checks below)
    tr.step(torch.from_numpy(data['rgb']).cuda(), torch.from_numpy(data['labels']).cuda())
    torch.cuda.synchronize()
    return net, tr, w, data


def _conv_backward_refs(x, dy, w_hwio, k):
    """(dW [HWIO], db, dx) of y = conv_same(x, w) + b for the given output gradient (torch-CPU fp32)."""
    wt = _w_oihw(w_hwio)
    pad = (k - 1) // 2
    dw = torch.nn.grad.conv2d_weight(x, wt.shape, dy, padding=pad).permute(2, 3, 1, 0).numpy()
    dx = torch.nn.grad.conv2d_input(x.shape, wt, dy, padding=pad)
    return dw, dy.sum((0, 2, 3)).numpy(), dx


def _check_trunk(L, Gm, gradview, kernel_of, x0, worst, tag=''):
    """The 13 trunk convs of one modality on the step's own state.  L: forward maps by layer name; Gm: gradient maps by
    the trainer's tags; gradview(layer, kind): the flat gradient buffer's view; kernel_of(layer): the step's fp32 HWIO
    kernel (before the update); x0: the raw input (NCHW).  Returns the gradient map of every conv's output."""
    from modular_semantic_segmentation_amd.fcn import ENCODER
    names = [nm for nm, _, _ in ENCODER]
    pool_after = {nm: pl for nm, _, pl in ENCODER}
    gout = {}
    for nm in names:
        gout[nm] = Gm['g_' + nm] if (nm == 'conv4_3' or not pool_after[nm]) else Gm['r_' + nm]
    prev = None
    for nm in names:
        xin = prev
        prev = pool_after[nm] if pool_after[nm] else nm
        if nm == 'conv1_1':
            continue
        x, dy = _nchw(L[xin]), _nchw(gout[nm])
        wq = fo.round_bf16(kernel_of(nm))                  # the step's forward / dgrad weights
        dw, db, dx = _conv_backward_refs(x, dy, wq, 3)
        worst[tag + 'dW ' + nm] = _close(gradview(nm, 'kernel').cpu().numpy(), dw, 1e-3, tag + 'dW ' + nm)
        worst[tag + 'db ' + nm] = _close(gradview(nm, 'bias').cpu().numpy(), db, 1e-3, tag + 'db ' + nm)
        if xin.startswith('pool'):
            # Conv2DBackpropInput -> gradient of the pooled map; MaxPoolGrad + ReluGrad route it to the conv above
            worst[tag + 'dx ' + nm] = _close(_nchw(Gm['g_' + xin]), fo.round_bf16(dx), 1e-2, tag + 'dx ' + nm)
            above = names[names.index(nm) - 1]
            y = _nchw(L[above])
            dpool = _nchw(Gm['g_' + xin])
            _, idx = F.max_pool2d(y, 2, 2, return_indices=True)      # first maximum of every window, like the kernel
            routed = torch.zeros_like(y).flatten(2).scatter_(2, idx.flatten(2), dpool.flatten(2)).view_as(y)
            routed = routed * (y > 0)
            assert torch.equal(_nchw(Gm['r_' + above]), routed), tag + 'pool route of ' + above
        else:
            ref = fo.round_bf16(dx * (_nchw(L[xin]) > 0))
            worst[tag + 'dx ' + nm] = _close(_nchw(gout[xin]), ref, 1e-2, tag + 'dx ' + nm)
    # conv1_1 (fp32 first layer): filter gradient from the raw image
    dw, db, _ = _conv_backward_refs(x0, _nchw(gout['conv1_1']), kernel_of('conv1_1'), 3)
    worst[tag + 'dW conv1_1'] = _close(gradview('conv1_1', 'kernel').cpu().numpy(), dw, 1e-3, tag + 'dW conv1_1')
    worst[tag + 'db conv1_1'] = _close(gradview('conv1_1', 'bias').cpu().numpy(), db, 1e-3, tag + 'db conv1_1')
    return gout


def test_plain_training_step_layer_by_layer(ops, tmp_path):
    """FcnTrainer (no batch norm): conv -> relu [-> pool] chain, second path into conv4_3 through score_conv4."""
    net, tr, w, data = _setup(False, tmp_path)
    e = tr.e
    L = {key[0]: act for key, act in e._arena.items() if isinstance(key[0], str) and hasattr(act, 'interior')}
    Gm = {key[0]: act for key, act in tr._g.items() if hasattr(act, 'interior')}
    worst = {}
    x0 = torch.from_numpy(data['rgb']).permute(0, 3, 1, 2).contiguous()
    _check_trunk(L, Gm, lambda nm, kind: tr.view(tr.grad, nm, kind), lambda nm: w['rgb/%s/kernel' % nm], x0, worst)
    # the 1x1 score convs (padded to 64 units) and the AddN into conv4_3
    for nm, src, g in (('score_conv4', 'conv4_3', 'ds4'), ('score_conv5', 'conv5_3', 'ds5')):
        kp = np.zeros((1, 1, 512, e.Up), np.float32)
        kp[..., :U] = w['rgb/%s/kernel' % nm]
        dw, db, dx = _conv_backward_refs(_nchw(L[src]), _nchw(Gm[g]), fo.round_bf16(kp), 1)
        worst['dW ' + nm] = _close(tr.view(tr.grad, nm, 'kernel').cpu().numpy(), dw, 1e-3, 'dW ' + nm)
        worst['db ' + nm] = _close(tr.view(tr.grad, nm, 'bias').cpu().numpy(), db, 1e-3, 'db ' + nm)
        if nm == 'score_conv5':
            ref = fo.round_bf16(dx * (_nchw(L['conv5_3']) > 0))
            worst['dx ' + nm] = _close(_nchw(Gm['g_conv5_3']), ref, 1e-2, 'dx ' + nm)
        else:
            ref = fo.round_bf16((dx + _nchw(Gm['r_conv4_3'])) * (_nchw(L['conv4_3']) > 0))       # AddN, then ReluGrad
            worst['dx ' + nm] = _close(_nchw(Gm['g_conv4_3']), ref, 1e-2, 'dx score_conv4 + pool4 route')
    # bilinear x2 deconv + add: ds4 = dfused * (s4 > 0); ds5 = relu-masked transpose of the x2 upsampling
    dfused = _nchw(Gm['dfused'])
    assert torch.equal(_nchw(Gm['ds4']), dfused * (_nchw(L['score_conv4']) > 0))
    k4 = torch.from_numpy(fo.bilinear_kernel(4, e.Up)).permute(3, 2, 0, 1).contiguous()
    s5 = _nchw(L['score_conv5']).requires_grad_(True)
    up = F.relu(F.conv_transpose2d(s5, k4, stride=2, padding=1))
    up.backward(dfused)
    ref = fo.round_bf16((s5.grad * (s5.detach() > 0)))
    worst['ds5'] = _close(_nchw(Gm['ds5']), ref, 1e-2, 'ds5')
    print('worst relative errors:', {k: float('%.2g' % v) for k, v in sorted(worst.items(), key=lambda kv: -kv[1])[:6]})
    assert len(worst) == 12 * 3 + 2 * 3 + 1 + 2


def test_joint_model_training_step_layer_by_layer(ops, tmp_path):
    """FusionFcnTrainer (fusion_fcn.py:11-40,50-92): both trunks (the walk shared with FcnTrainer, the gradient maps tagged
    per modality), the two fused 1x1 convs over the channel concat taken block by block -- filter gradient per 512-row
    block, data gradient per modality with that block's weights (+ the pool4 route: AddN) -- on the step's own state."""
    from modular_semantic_segmentation_amd import get_model
    rng = np.random.default_rng(4)
    data = {'rgb': rng.integers(0, 256, (2, H, W, 3)).astype(np.float32),
            'depth': rng.integers(0, 65536, (2, H, W, 1)).astype(np.float32),
            'labels': rng.integers(-1, C, (2, H, W)).astype(np.int32)}
    prefixes, nch = {'rgb': 'rgb', 'depth': 'depth'}, {'rgb': 3, 'depth': 1}
    w = fo.init_fusion_fcn_weights(prefixes, nch, U, C, seed=2, bias_scale=0.02)
    w['rgb_conv1_1/kernel'] *= 0.02
    w['depth_conv1_1/kernel'] *= 1e-4
    for k in w:
        if k.endswith('/kernel') and 'upscore' not in k and 'conv1_1' not in k:
            w[k] *= 1.6
    net = get_model('fusion_fcn')(prefixes, nch, U, C, trainer='rmsprop', learning_rate=1e-3, batchsize=2, seed=5)
    net.variables.update(w)
    net._variables_changed()
    tr = net._ensure_trainer()
    tr.step({m: torch.from_numpy(data[m]).cuda() for m in prefixes}, torch.from_numpy(data['labels']).cuda())
    torch.cuda.synchronize()
    A = {key[0]: act for key, act in tr._a.items() if hasattr(act, 'interior')}
    worst = {}
    ds4, ds5 = _nchw(A['ds4']), _nchw(A['ds5'])
    for i, m in enumerate(('rgb', 'depth')):
        trunk = tr.e.trunks[m]
        L = {key[0]: act for key, act in trunk._arena.items() if hasattr(act, 'interior')}
        Gm = {tag[:-len(m) - 1]: act for tag, act in A.items() if tag.endswith('_' + m)}
        Gm['g_conv5_3'] = A['g5_' + m]                    # the data gradient of the fused 1x1 conv, per modality
        x0 = torch.from_numpy(data[m]).permute(0, 3, 1, 2).contiguous()
        _check_trunk(L, Gm, lambda nm, kind, m=m: tr.view(tr.grad, (m, nm), kind), lambda nm, m=m: w['%s_%s/kernel' % (m, nm)],
                     x0, worst, tag=m + ' ')
        rows = slice(512 * i, 512 * (i + 1))
        for nm, src, g in (('fused_score_conv4', 'conv4_3', ds4), ('fused_score_conv5', 'conv5_3', ds5)):
            kp = np.zeros((1, 1, 512, tr.e.Up), np.float32)
            kp[..., :U] = w[nm + '/kernel'][:, :, rows, :]
            dw, db, dx = _conv_backward_refs(_nchw(L[src]), g, fo.round_bf16(kp), 1)
            got = tr.view(tr.grad, nm, 'kernel')[:, :, rows, :].cpu().numpy()
            worst['%s dW %s' % (m, nm)] = _close(got, dw, 1e-3, '%s dW %s' % (m, nm))
            if i == 0:
                worst['db ' + nm] = _close(tr.view(tr.grad, nm, 'bias').cpu().numpy(), db, 1e-3, 'db ' + nm)
            if src == 'conv5_3':
                ref = fo.round_bf16(dx * (_nchw(L['conv5_3']) > 0))
                worst['%s dx %s' % (m, nm)] = _close(_nchw(A['g5_' + m]), ref, 1e-2, '%s dx %s' % (m, nm))
            else:
                ref = fo.round_bf16((dx + _nchw(Gm['r_conv4_3'])) * (_nchw(L['conv4_3']) > 0))
                worst['%s dx %s' % (m, nm)] = _close(_nchw(Gm['g_conv4_3']), ref, 1e-2, '%s dx %s + pool4 route' % (m, nm))
    print('worst relative errors:', {k: float('%.2g' % v) for k, v in sorted(worst.items(), key=lambda kv: -kv[1])[:6]})
    assert len(worst) == 2 * (12 * 3 + 2) + 2 * 4 + 2


def test_batch_norm_training_step_layer_by_layer(ops, tmp_path):
    """FcnBnTrainer: conv -> z -> batch norm (batch statistics) -> relu: per layer the batch-norm backward (dz from dy
    with the stored z / y and the step's own statistics; d gamma, d beta) and the conv gradients on (stored input, dz)."""
    from modular_semantic_segmentation_amd.fcn import ENCODER
    net, tr, w, data = _setup(True, tmp_path)
    A = {key[0]: act for key, act in tr._a.items() if hasattr(act, 'interior')}
    names = [nm for nm, _, _ in ENCODER]
    pool_after = {nm: pl for nm, _, pl in ENCODER}
    worst = {}
    prev = None
    for nm in names:
        xin_tag = None if prev is None else (prev if prev.startswith('pool') else 'y_' + prev)
        prev = pool_after[nm] if pool_after[nm] else nm
        z, y, dz = _nchw(A['z_' + nm]), _nchw(A['y_' + nm]), _nchw(A['dz_' + nm])
        # the gradient w.r.t. y that the step fed into this batch norm
        if nm == 'conv5_3':
            dy = _nchw(A['g_conv5_3'])
        elif nm == 'conv4_3':
            dy = _nchw(A['g_conv4_3'])
        elif pool_after[nm]:
            # MaxPoolGrad happens INSIDE the batch-norm gradient (ops.bn_pool_backward): route the pooled gradient here, to
            # the first maximum of each 2x2 window of the stored activations if it is positive
            dpool = _nchw(A['dx_' + names[names.index(nm) + 1]])
            n_, c_, h_, w_ = y.shape
            win = y.reshape(n_, c_, h_ // 2, 2, w_ // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(n_, c_, h_ // 2, w_ // 2, 4)
            best = win.argmax(-1)                                           # (first occurrence on ties)
            sel = torch.nn.functional.one_hot(best, 4).to(y.dtype) * (win.max(-1).values > 0).unsqueeze(-1) * dpool.unsqueeze(-1)
            dy = sel.reshape(n_, c_, h_ // 2, w_ // 2, 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(n_, c_, h_, w_)
        else:
            dy = _nchw(A['dx_' + names[names.index(nm) + 1]])
        # [TF1] batch norm backward with the relu mask, biased batch statistics of the stored z (fp32)
        g = dy * (y > 0)
        m = z.shape[0] * z.shape[2] * z.shape[3]
        mean = z.mean((0, 2, 3), keepdim=True)
        var = ((z - mean) ** 2).mean((0, 2, 3), keepdim=True)
        inv = 1.0 / torch.sqrt(var + 1e-3)
        zh = (z - mean) * inv
        gamma = torch.from_numpy(np.ones(z.shape[1], np.float32)).view(1, -1, 1, 1)        # [TF1] initial gamma
        dgamma, dbeta = (g * zh).sum((0, 2, 3)), g.sum((0, 2, 3))
        ref = gamma * inv * (g - dbeta.view(1, -1, 1, 1) / m - zh * dgamma.view(1, -1, 1, 1) / m)
        worst['dz ' + nm] = _close(dz, fo.round_bf16(ref), 1e-2, 'batch-norm dz ' + nm)
        worst['dgamma ' + nm] = _close(tr.view(tr.grad, nm, 'gamma').cpu().numpy(), dgamma.numpy(), 2e-3, 'dgamma ' + nm)
        worst['dbeta ' + nm] = _close(tr.view(tr.grad, nm, 'beta').cpu().numpy(), dbeta.numpy(), 2e-3, 'dbeta ' + nm)
        if nm == 'conv1_1':
            x0 = torch.from_numpy(data['rgb']).permute(0, 3, 1, 2).contiguous()
            dw, db, _ = _conv_backward_refs(x0, dz, w['rgb/conv1_1/kernel'], 3)
        else:
            dw, db, dx = _conv_backward_refs(_nchw(A[xin_tag]), dz, fo.round_bf16(w['rgb/%s/kernel' % nm]), 3)
            worst['dx ' + nm] = _close(_nchw(A['dx_' + nm]), fo.round_bf16(dx), 1e-2, 'dx ' + nm)
        worst['dW ' + nm] = _close(tr.view(tr.grad, nm, 'kernel').cpu().numpy(), dw, 1e-3, 'dW ' + nm)
        worst['db ' + nm] = _close(tr.view(tr.grad, nm, 'bias').cpu().numpy(), db, 1e-3, 'db ' + nm)
    print('worst relative errors:', {k: float('%.2g' % v) for k, v in sorted(worst.items(), key=lambda kv: -kv[1])[:6]})
    assert len(worst) == 13 * 5 + 12


SHALLOW = [('block_layer_1', 'a', (64, 128, 1, True)), ('block_layer_2', 'a', (64, 128, 1, False)),
           ('block_layer_4', 'a', (64, 256, 2, True)), ('block_layer_7', 'b', (64, 64, 256, 1, 2, False)),
           ('block_layer_8', 'a', (128, 512, 2, True)), ('block_layer_14', 'b', (128, 128, 512, 2, 4, False))]


def test_adapnet_training_step_op_by_op(ops, monkeypatch):
    """AdapnetTrainer on the 6-block graph with every op type (adapnet.py:12-173, loss / optimizer :190-203): its backward
    is a tape over a DAG, so instead of walking named buffers the test RECORDS every filter-gradient, data-gradient and
    batch-norm-backward call of the step -- the exact operand maps as they sit in HBM at call time -- and recomputes each
    one with torch-CPU fp32 on the spot (the kernel a data gradient used is identified by its packed buffer).  Same
    tolerances as the layer-by-layer tests above; ~25 convs (1x1, 3x3, the 7x7 stride-2 conv as a 3x3 over 9 gathered
    groups, the stacked atrous pair as one 1x1) and ~30 batch norms are covered."""
    from modular_semantic_segmentation_amd.adapnet import AdapnetEngine
    from modular_semantic_segmentation_amd.adapnet_trainer import AdapnetTrainer
    from oracle import adapnet_oracle as ao
    h, w = 64, 96
    rng = np.random.default_rng(0)
    x = rng.integers(0, 256, (2, h, w, 3)).astype(np.float32)
    labels = rng.integers(-1, C, (2, h, w)).astype(np.int32)
    w_ = ao.init_adapnet_weights('rgb', 3, U, C, seed=1, gain=1.3, blocks=SHALLOW)
    w_['rgb/block_0_1/kernel'] *= 0.02
    eng = AdapnetEngine('rgb', 3, U, C, w_, blocks=SHALLOW)
    tr = AdapnetTrainer(eng, 'rmsprop', 1e-3)
    tr.load_from_variables(w_)
    torch.cuda.synchronize()
    # fp32 kernels (as the MFMA convs consume them, rounded to bf16) of every packed data-gradient buffer, before the update
    kernels = {}
    for key, buf in tr.wd.items():
        name = key[:-len('/stage_2')] if key.endswith('/stage_2') else None
        is_pair = name is not None and any(b[0] == name and b[1] == 'b' for b in SHALLOW)
        kern = tr._pair_kernel(name) if is_pair else tr._kernel_for(key)
        kernels[buf.data_ptr()] = fo.round_bf16(kern.detach().float().cpu().numpy().copy())
    worst = {'wgrad': 0.0, 'dbias': 0.0, 'dgrad': 0.0, 'bn_dz': 0.0, 'bn_dgamma': 0.0, 'bn_dbeta': 0.0}
    calls = {'wgrad': 0, 'dgrad': 0, 'bn': 0}

    def rel(got, ref):
        got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
        return float(np.abs(got - ref).max() / (np.abs(ref).max() + 1e-30))

    orig_wgrad, orig_dgrad, orig_bn = ops.conv2d_bwd_filter, ops.conv2d_bwd_data, ops.bn_backward

    def wgrad(xa, dy, dw, dbias, k, workspace=None):
        before = dw.detach().clone()
        bias_before = dbias.detach().clone() if dbias is not None else None
        orig_wgrad(xa, dy, dw, dbias, k, workspace=workspace)
        torch.cuda.synchronize()
        xr, dyr = _nchw(xa), _nchw(dy)
        ref = torch.nn.grad.conv2d_weight(xr, (dy.c, xa.c, k, k), dyr, padding=(k - 1) // 2).permute(2, 3, 1, 0).numpy()
        worst['wgrad'] = max(worst['wgrad'], rel((dw - before).cpu().numpy(), ref))
        if dbias is not None:
            worst['dbias'] = max(worst['dbias'], rel((dbias - bias_before).cpu().numpy(), dyr.sum((0, 2, 3)).numpy()))
        calls['wgrad'] += 1

    def dgrad(dy, w_packed_dgrad, zero_bias, dx, k, relu_ref=None, addend=None):
        dyr = _nchw(dy)
        add = _nchw(addend) if addend is not None else None
        out = orig_dgrad(dy, w_packed_dgrad, zero_bias, dx, k, relu_ref=relu_ref, addend=addend)
        torch.cuda.synchronize()
        wt = _w_oihw(kernels[w_packed_dgrad.data_ptr()])
        ref = torch.nn.grad.conv2d_input((dx.n, dx.c, dx.h, dx.w), wt, dyr, padding=(k - 1) // 2)
        if add is not None:
            ref = ref + add
        if relu_ref is not None:
            ref = ref * (_nchw(relu_ref) > 0)
        worst['dgrad'] = max(worst['dgrad'], rel(_nchw(dx).numpy(), fo.round_bf16(ref).numpy()))
        calls['dgrad'] += 1
        return out

    def bn_backward(dy, y, z, gamma, st, dgamma, dbeta, dz, sync=False):
        dyr, zr = _nchw(dy), _nchw(z)                     # before the call: dz may alias dy
        yr = _nchw(y) if y is not None else None
        g0, b0 = dgamma.detach().clone(), dbeta.detach().clone()
        out = orig_bn(dy, y, z, gamma, st, dgamma, dbeta, dz, sync=sync)
        torch.cuda.synchronize()
        g = dyr * (yr > 0) if yr is not None else dyr
        m = zr.shape[0] * zr.shape[2] * zr.shape[3]
        mean = zr.mean((0, 2, 3), keepdim=True)
        var = ((zr - mean) ** 2).mean((0, 2, 3), keepdim=True)
        inv = 1.0 / torch.sqrt(var + 1e-3)
        zh = (zr - mean) * inv
        dga, dbe = (g * zh).sum((0, 2, 3)), g.sum((0, 2, 3))
        ref = gamma.detach().float().cpu().view(1, -1, 1, 1) * inv * (g - dbe.view(1, -1, 1, 1) / m - zh * dga.view(1, -1, 1, 1) / m)
        worst['bn_dz'] = max(worst['bn_dz'], rel(_nchw(dz).numpy(), fo.round_bf16(ref).numpy()))
        worst['bn_dgamma'] = max(worst['bn_dgamma'], rel((dgamma - g0).cpu().numpy(), dga.numpy()))
        worst['bn_dbeta'] = max(worst['bn_dbeta'], rel((dbeta - b0).cpu().numpy(), dbe.numpy()))
        calls['bn'] += 1
        return out

    monkeypatch.setattr(ops, 'conv2d_bwd_filter', wgrad)
    monkeypatch.setattr(ops, 'conv2d_bwd_data', dgrad)
    monkeypatch.setattr(ops, 'bn_backward', bn_backward)
    tr.step(torch.from_numpy(x).cuda(), torch.from_numpy(labels).cuda())
    torch.cuda.synchronize()
    print('adapnet step: calls', calls, 'worst relative errors', {k: float('%.2g' % v) for k, v in worst.items()})
    assert calls['wgrad'] >= 22 and calls['dgrad'] >= 20 and calls['bn'] >= 24
    assert worst['wgrad'] < 1e-3 and worst['dbias'] < 1e-3
    assert worst['dgrad'] < 1e-2 and worst['bn_dz'] < 1e-2
    assert worst['bn_dgamma'] < 2e-3 and worst['bn_dbeta'] < 2e-3


def test_adapnet_deconv_kernel_gradients_on_frozen_state(ops, monkeypatch):
    """AdapNet TRAINS its two transposed-conv kernels (adapnet.py:155-163 calls custom_layers.deconv2d:71-121 without
    trainable=False).  Both kernel gradients and both data gradients of one step are recomputed by torch-CPU autograd
    through F.conv_transpose2d itself -- from the exact bf16 input maps and upstream gradients the step left in HBM --
    so the whole chain (space-to-depth shuffle, 3x3 filter / data gradient on the derived phase kernel, gather back
    through the inverse index map) is pinned against the operation the reference differentiates.  A second step from
    DENSE kernels (what a trained checkpoint holds) repeats it."""
    from modular_semantic_segmentation_amd.adapnet import AdapnetEngine
    from modular_semantic_segmentation_amd.adapnet_trainer import AdapnetTrainer
    from oracle import adapnet_oracle as ao
    h, w = 64, 96
    rng = np.random.default_rng(0)
    x = torch.from_numpy(rng.integers(0, 256, (2, h, w, 3)).astype(np.float32)).cuda()
    labels = torch.from_numpy(rng.integers(-1, C, (2, h, w)).astype(np.int32)).cuda()
    w_ = ao.init_adapnet_weights('rgb', 3, U, C, seed=1, gain=1.3, blocks=SHALLOW)
    w_['rgb/block_0_1/kernel'] *= 0.02
    eng = AdapnetEngine('rgb', 3, U, C, w_, blocks=SHALLOW)
    tr = AdapnetTrainer(eng, 'rmsprop', 1e-3)
    tr.load_from_variables(w_)
    cap = {}
    orig_s2d, orig_s2dd, orig_dgrad = ops.space_to_depth, ops.space_to_depth_dense, ops.conv2d_bwd_data

    def s2d(g, stride, out=None):
        cap['g1'] = _nchw(g)
        res = orig_s2d(g, stride, out)
        torch.cuda.synchronize()
        # the shuffle itself: phase channel (py*s + px)*C + c  <->  pixel (s*qy + py, s*qx + px)
        n, c, hh, ww = cap['g1'].shape
        want = cap['g1'].view(n, c, hh // stride, stride, ww // stride, stride).permute(0, 3, 5, 1, 2, 4).reshape(
            n, stride * stride * c, hh // stride, ww // stride)
        assert torch.equal(_nchw(res), want)
        assert not res.t[:, 0].any() and not res.t[:, :, -1].any()
        return res

    def s2dd(g, stride, out):
        cap['g2'] = g.detach().float().cpu().permute(0, 3, 1, 2).contiguous()
        res = orig_s2dd(g, stride, out)
        torch.cuda.synchronize()
        n, c, hh, ww = cap['g2'].shape
        cp = out.c // (stride * stride)
        gp = F.pad(cap['g2'], (0, 0, 0, 0, 0, cp - c))
        want = gp.view(n, cp, hh // stride, stride, ww // stride, stride).permute(0, 3, 5, 1, 2, 4).reshape(
            n, stride * stride * cp, hh // stride, ww // stride)
        assert torch.equal(_nchw(res), fo.round_bf16(want))
        return res

    def dgrad(dy, w_packed_dgrad, zero_bias, dx, k, relu_ref=None, addend=None):
        out = orig_dgrad(dy, w_packed_dgrad, zero_bias, dx, k, relu_ref=relu_ref, addend=addend)
        for scope in ('first_deconvolution_upconv', 'second_deconvolution_upconv'):
            if w_packed_dgrad.data_ptr() == tr.wd[scope].data_ptr():
                torch.cuda.synchronize()
                cap['dx_' + scope] = _nchw(dx)
        return out

    monkeypatch.setattr(ops, 'space_to_depth', s2d)
    monkeypatch.setattr(ops, 'space_to_depth_dense', s2dd)
    monkeypatch.setattr(ops, 'conv2d_bwd_data', dgrad)
    for step in range(2):
        kern = {s: tr.view(tr.param, s, 'kernel').detach().float().cpu().numpy().copy() for s in tr.deconv_shape}
        if step == 1:
            for s in kern:      # after one optimizer step the kernels are dense: what a trained checkpoint holds
                f, cin = tr.deconv_real[s]
                diag = np.zeros(kern[s].shape[2:], bool)
                diag[np.arange(min(f, cin)), np.arange(min(f, cin))] = True
                assert np.abs(kern[s][:, :, ~diag]).max() > 0
        tr.step(x, labels)
        torch.cuda.synchronize()
        n = 2
        ins = {'first_deconvolution_upconv': tr._a[('y_deconv_in', n, h // 16, w // 16, tr.width)],
               'second_deconvolution_upconv': tr._a[('merge', n, h // 8, w // 8, eng.Up)]}
        for scope, stride, gkey in (('first_deconvolution_upconv', 2, 'g1'), ('second_deconvolution_upconv', 8, 'g2')):
            f, cin = tr.deconv_real[scope]
            k = kern[scope].shape[0]
            xin = _nchw(ins[scope])[:, :cin].clone().requires_grad_(True)
            wt = torch.from_numpy(fo.round_bf16(kern[scope][:, :, :f, :cin])).requires_grad_(True)
            y = F.conv_transpose2d(xin, wt.permute(3, 2, 0, 1), stride=stride, padding=(k - stride) // 2)
            y.backward(fo.round_bf16(cap[gkey][:, :f]))             # the upstream gradient as the MFMA convs read it (bf16)
            got = tr.view(tr.grad, scope, 'kernel').detach().float().cpu().numpy()
            _close(got[:, :, :f, :cin], wt.grad.numpy(), 2e-3, 'step %d: d %s/kernel' % (step, scope))
            if f < got.shape[2] or cin < got.shape[3]:         # channel padding: exactly zero gradient, stays zero
                mask = np.ones(got.shape[2:], bool)
                mask[:f, :cin] = False
                assert not got[:, :, mask].any()
            _close(cap['dx_' + scope][:, :cin].numpy(), fo.round_bf16(xin.grad).numpy(), 1e-2,
                   'step %d: data gradient of %s' % (step, scope))
