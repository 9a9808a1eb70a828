"""Host-side counterparts of xview/models/custom_layers.py that the MI355X path needs:
the bilinear transposed-conv constant and the check that an imported deconv kernel IS that
constant (the HIP path applies it as a depthwise 2x2-tap interpolation)."""
import numpy as np


def bilinear_filter(filter_shape):
    """Bilinear transposed-conv kernel, layout [kh, kw, out, in], zero off the channel diagonal
    (reference: custom_layers.py:8-25 bilinear_filter_initializer)."""
    kh, kw, cout, cin = filter_shape
    factor = np.ceil(kh / 2.0)
    center = (2 * factor - 1 - factor % 2) / (2.0 * factor)
    w1 = 1 - np.abs(np.arange(kh) / factor - center)
    w2 = 1 - np.abs(np.arange(kw) / factor - center)
    weights = np.zeros(filter_shape)
    idx = np.arange(min(cout, cin))
    weights[:, :, idx, idx] = np.outer(w1, w2)[:, :, None]
    return weights.astype(np.float32)


def is_bilinear_filter(kernel, atol=1e-6):
    """True if `kernel` ([k,k,U,U]) equals the constant the reference initialises and never trains
    (simple_fcn.py:80-83,117-130: deconvs are always `trainable=False`)."""
    kernel = np.asarray(kernel)
    if kernel.ndim != 4 or kernel.shape[0] != kernel.shape[1] or kernel.shape[2] != kernel.shape[3]:
        return False
    return bool(np.allclose(kernel, bilinear_filter(kernel.shape), rtol=0, atol=atol))


def dense_deconv_as_conv3x3(kernel, stride):
    """Re-arrange a dense transposed-conv kernel [k, k, out, in] (tf.layers.conv2d_transpose layout,
    custom_layers.py:71-121) with k = 2 * stride and 'same' padding into the HWIO kernel [3, 3, in, s*s*out] of a
    stride-1 3x3 'same' convolution that computes all s*s output phases at the input resolution:

        [TF1] out[s*q + p] = sum_i in[i] * W[p + pad - (i - q)*s],  pad = (k - s) // 2 = s / 2
        =>  phase p sees in[q + d] for d in {-1, 0, 1} through tap ky = p + pad - d*s (when 0 <= ky < k)

    so K[d_y + 1, d_x + 1, ci, (p_y*s + p_x)*out + co] = W[ky, kx, co, ci].  xv_deconv_dense_fwd runs it on the MFMA
    conv and un-shuffles the phases (depth-to-space)."""
    kernel = np.asarray(kernel, np.float32)
    k, k2, cout, cin = kernel.shape
    s = int(stride)
    if k != k2 or k != 2 * s:
        raise NotImplementedError('dense transposed conv: kernel %dx%d with stride %d (only k = 2*stride)' % (k, k2, s))
    pad = (k - s) // 2
    out = np.zeros((3, 3, cin, s * s * cout), np.float32)
    for py in range(s):
        for dy in (-1, 0, 1):
            ky = py + pad - dy * s
            if not 0 <= ky < k:
                continue
            for px in range(s):
                for dx in (-1, 0, 1):
                    kx = px + pad - dx * s
                    if 0 <= kx < k:
                        out[dy + 1, dx + 1, :, (py * s + px) * cout:(py * s + px + 1) * cout] = kernel[ky, kx].T
    return out


# =====================================================================================================================
# The reference's L1 layer functions under their own names (custom_layers.py:8-25,71-139) over the HIP kernels.
#
# Tensors: `inputs` is an ops.Act (bf16 padded NHWC, what every layer here returns) or a dense float32 [N,H,W,C] device
# tensor.  Variables: TensorFlow keeps them in the graph under `<scope>/<name>/...`; here the caller passes the dict
# (`variables=`, reference npz schema: numpy arrays or device tensors) and the scope (`scope=`, the model prefix).  Keyword
# arguments TensorFlow needs and this path does not (`reuse`, `trainable`, initialisers, regularisers, `data_format=
# 'channels_last'`) are accepted and ignored.  Packed weights are cached per (variables dict, layer): call
# `clear_layer_cache()` after changing a dict in place.
# =====================================================================================================================
_LAYER_CACHE = {}         # key (kind, id(variables), ...) -> (variables, payload): the entry keeps the dict alive, so that the
#                           id in its key cannot be handed to another dict while the entry exists
_IGNORED = ('reuse', 'trainable', 'kernel_initializer', 'bias_initializer', 'kernel_regularizer', 'bias_regularizer',
            'activity_regularizer', 'data_format')
BN_EPSILON = 1e-3       # [TF1] tf.layers.batch_normalization default


def clear_layer_cache():
    _LAYER_CACHE.clear()


def _cache_get(key, variables):
    ent = _LAYER_CACHE.get(key)
    return ent[1] if ent is not None and ent[0] is variables else None


def _cache_put(key, variables, payload):
    _LAYER_CACHE[key] = (variables, payload)
    return payload


class _ConstantInitializer(object):
    """What tf.constant_initializer(value, verify_shape=True) is to its callers: called with a shape it hands the value out."""

    def __init__(self, value):
        self.value = value

    def __call__(self, shape=None, dtype=None, partition_info=None):
        if shape is not None and tuple(shape) != self.value.shape:
            raise ValueError('bilinear_filter_initializer: shape %s requested, the constant has %s' % (tuple(shape), self.value.shape))
        return self.value.astype(dtype or np.float32)


def bilinear_filter_initializer(filter_shape):
    """custom_layers.py:8-25: an initializer whose value is the bilinear transposed-conv kernel [k, k, out, in]."""
    return _ConstantInitializer(bilinear_filter(list(filter_shape)))


def _is_relu(activation):
    if activation is None:
        return False
    name = activation if isinstance(activation, str) else getattr(activation, '__name__', '')
    if name not in ('relu', 'relu_'):
        raise NotImplementedError('activation %r (the FCN path uses tf.nn.relu or none)' % (activation,))
    return True


def _square(v, what):
    if isinstance(v, (list, tuple)):
        if len(v) != 2 or v[0] != v[1]:
            raise NotImplementedError('%s %r (square only)' % (what, v))
        return int(v[0])
    return int(v)


def _var(variables, scope, name, kind, device=None):
    key = '%s/%s/%s' % (scope, name, kind) if scope else '%s/%s' % (name, kind)
    if variables is None or key not in variables:
        raise KeyError('variable %r (pass variables=<reference-schema dict>, scope=<model prefix>, name=<layer>)' % key)
    v = variables[key]
    if hasattr(v, 'detach'):
        v = v.detach().cpu().numpy()
    return np.asarray(v, np.float32), key


def _as_act(inputs):
    from . import ops
    if isinstance(inputs, ops.Act):
        return inputs
    return ops.Act.from_dense(inputs)


def _bn_affine(variables, scope, name, channels):
    """Inference batch norm as y = x * s + t from gamma / beta / moving_mean / moving_variance."""
    g, _ = _var(variables, scope, name, 'gamma')
    b, _ = _var(variables, scope, name, 'beta')
    mm, _ = _var(variables, scope, name, 'moving_mean')
    mv, _ = _var(variables, scope, name, 'moving_variance')
    s = g / np.sqrt(mv + BN_EPSILON)
    return s.astype(np.float32), (b - mm * s).astype(np.float32)


def _bn_training_state(variables, scope, name, channels, real, device):
    """Device copies of gamma / beta / moving statistics padded to `channels` lanes + the BnState scratch, cached; after the
    pass `_bn_write_back` stores the updated moving statistics in the dict (the reference's UPDATE_OPS)."""
    import torch
    from . import ops
    key = ('bn', id(variables), scope, name, channels, str(device))
    st = _cache_get(key, variables)
    if st is None:
        def pad(kind, fill):
            v, _ = _var(variables, scope, name, kind)
            out = np.full(channels, fill, np.float32)
            out[:real] = v
            return torch.from_numpy(out).to(device)
        st = {'gamma': pad('gamma', 1.0), 'beta': pad('beta', 0.0), 'mm': pad('moving_mean', 0.0),
              'mv': pad('moving_variance', 1.0), 'st': ops.BnState(channels, device)}
        _cache_put(key, variables, st)
    return st


def _bn_write_back(variables, scope, name, st, real):
    pre = '%s/%s/' % (scope, name) if scope else name + '/'
    variables[pre + 'moving_mean'] = st['mm'][:real].cpu().numpy().copy()
    variables[pre + 'moving_variance'] = st['mv'][:real].cpu().numpy().copy()


def conv2d(inputs, filters, kernel_size, batch_normalization=False, training=False, variables=None, scope='', **kwargs):
    """tf.layers.conv2d with batch norm between the conv and its activation (custom_layers.py:124-139) on the MFMA conv
    kernels.  kwargs as the reference passes them: name, activation (tf.nn.relu -> 'relu' / torch.relu / None),
    padding='same', strides 1, use_bias.  Shapes the kernels take: 3x3 with 1..4 input channels and 64 filters (the first
    layer: fp32 operands), 1x1 / 3x3 with input channels a multiple of 64 (filters are padded to 64 lanes of zeros: the
    returned Act has the padded width, `.filters` the real one), and 1x1 onto at most 32 classes from at most 256
    channels without activation (the `score` layer: returns a dense float32 [N,H,W,filters] tensor).
    training=True normalises with the batch statistics and updates the moving averages in `variables`."""
    import torch
    from . import ops
    for k in list(kwargs):
        if k in _IGNORED:
            kwargs.pop(k)
    name = kwargs.pop('name', None)
    relu = _is_relu(kwargs.pop('activation', None))
    padding = kwargs.pop('padding', 'valid')
    strides = _square(kwargs.pop('strides', 1), 'strides')
    use_bias = kwargs.pop('use_bias', True)
    if kwargs:
        raise TypeError('conv2d: unexpected arguments %s' % sorted(kwargs))
    k = _square(kernel_size, 'kernel_size')
    if name is None:
        raise ValueError('conv2d needs name= (the variable scope of the layer)')
    if strides != 1 or (k != 1 and str(padding).lower() != 'same') or k not in (1, 3):
        raise NotImplementedError('conv2d: %dx%d, strides %d, padding %r (1x1 / 3x3, stride 1, same)' % (k, k, strides, padding))
    dense_in = not isinstance(inputs, ops.Act)
    dev = inputs.device if dense_in else inputs.t.device
    cin = int(inputs.shape[-1]) if dense_in else inputs.c
    filters = int(filters)
    w, wkey = _var(variables, scope, name, 'kernel')
    if w.shape[:2] != (k, k) or w.shape[3] != filters or w.shape[2] > cin:
        raise ValueError('%s has shape %s, expected (%d, %d, <=%d, %d)' % (wkey, w.shape, k, k, cin, filters))
    b = _var(variables, scope, name, 'bias')[0] if use_bias else np.zeros(filters, np.float32)
    fold = batch_normalization and not training
    ckey = ('conv', id(variables), wkey, fold, cin, str(dev))
    ent = _cache_get(ckey, variables)
    first = cin <= 4
    score = (not first) and k == 1 and filters <= 32 and cin <= 256 and not relu
    cout_p = filters if (first or score) else (filters + 63) // 64 * 64
    if ent is None:
        wf, bf = w.astype(np.float64), b.astype(np.float64)
        if fold:
            s, t = _bn_affine(variables, scope, name, filters)
            wf, bf = wf * s, bf * s + t
        if w.shape[2] < cin:                                   # input lanes beyond the layer's channels are zero padding
            wf = np.concatenate([wf, np.zeros(w.shape[:2] + (cin - w.shape[2], filters))], 2)
        if first:
            if k != 3 or filters != 64:
                raise NotImplementedError('conv2d on %d input channels: 3x3 onto 64 filters only (conv1_1)' % cin)
            ent = (torch.from_numpy(wf.astype(np.float32)).to(dev), torch.from_numpy(bf.astype(np.float32)).to(dev))
        elif score:
            ent = (torch.from_numpy(wf.astype(np.float32).reshape(cin, filters)).to(dev).contiguous(),
                   torch.from_numpy(bf.astype(np.float32)).to(dev))
        else:
            if cin % 64:
                raise NotImplementedError('conv2d: %d input channels (a multiple of 64, or the first layer)' % cin)
            wp = np.zeros((k, k, cin, cout_p), np.float32)
            wp[..., :filters] = wf
            bp = np.zeros(cout_p, np.float32)
            bp[:filters] = bf
            ent = (ops.pack_conv_weights(torch.from_numpy(wp).to(dev)), torch.from_numpy(bp).to(dev))
        _cache_put(ckey, variables, ent)
    wdev, bdev = ent
    bn_train = batch_normalization and training
    if score:
        x = _as_act(inputs)
        out = torch.empty((x.n, x.h, x.w, filters), dtype=torch.float32, device=dev)
        ops.score_dense_fwd(x, wdev, bdev, filters, out)
        if bn_train:
            st = _bn_training_state(variables, scope, name, filters, filters, dev)
            ops.bn_dense_forward(out, st['gamma'], st['beta'], st['mm'], st['mv'], st['st'], out)
            _bn_write_back(variables, scope, name, st, filters)
        return out
    if first:
        x = inputs if dense_in else inputs.real()
        x = x.to(torch.float32).contiguous()
        y = ops.Act(x.shape[0], x.shape[1], x.shape[2], 64, dev)
        ops.conv2d_first_fwd(x, wdev, bdev, y, relu=relu and not bn_train)
    else:
        x = _as_act(inputs)
        y = ops.conv2d_fwd(x, wdev, bdev, k, relu=relu and not bn_train)[0]
    if bn_train:
        st = _bn_training_state(variables, scope, name, cout_p, filters, dev)
        ops.bn_forward(y, st['gamma'], st['beta'], st['mm'], st['mv'], st['st'], y, relu=relu)
        _bn_write_back(variables, scope, name, st, filters)
    y.filters = filters
    return y


def deconv2d(inputs, filters, kernel_size, strides=(1, 1), padding='valid', activation=None, use_bias=False, name=None,
             batch_normalization=True, training=False, variables=None, scope='', **kwargs):
    """custom_layers.py:71-121: tf.layers.conv2d_transpose initialised with the bilinear kernel, batch norm between the
    deconv and its activation.  The kernel is `variables['<scope>/<name>/kernel']` when the dict has one, else the bilinear
    constant.  A bilinear kernel with k = 2 * stride (x2, x8: what the FCN uses and never trains) runs as a depthwise
    four-tap interpolation; any other [k, k, out, in] kernel with k = 2 * stride as a 3x3 conv over the s*s output phases
    (xv_deconv_dense_fwd).  `filters` must equal the input's channel count for the bilinear form."""
    import torch
    from . import _lib, ops
    for k_ in list(kwargs):
        if k_ in _IGNORED:
            kwargs.pop(k_)
    if kwargs:
        raise TypeError('deconv2d: unexpected arguments %s' % sorted(kwargs))
    relu = _is_relu(activation)
    k, s = _square(kernel_size, 'kernel_size'), _square(strides, 'strides')
    if name is None:
        raise ValueError('deconv2d needs name=')
    if use_bias or str(padding).lower() != 'same' or k != 2 * s:
        raise NotImplementedError('deconv2d: kernel %d, strides %d, padding %r, use_bias %r (k = 2 * stride, same, no bias)'
                                  % (k, s, padding, use_bias))
    x = _as_act(inputs)
    dev = x.t.device
    filters = int(filters)
    real_in = getattr(x, 'filters', x.c)
    wkey = '%s/%s/kernel' % (scope, name) if scope else name + '/kernel'
    if variables is not None and wkey in variables:
        w = _var(variables, scope, name, 'kernel')[0]
    else:
        w = bilinear_filter([k, k, filters, real_in])
    if w.shape != (k, k, filters, real_in):
        raise ValueError('%s has shape %s, expected %s' % (wkey, w.shape, (k, k, filters, real_in)))
    cp = (filters + 63) // 64 * 64
    bn_train = batch_normalization and training
    scale = shift = None
    if batch_normalization and not training:
        sc, sh = _bn_affine(variables, scope, name, filters)
        sp, tp = np.ones(cp, np.float32), np.zeros(cp, np.float32)
        sp[:filters], tp[:filters] = sc, sh
        scale, shift = torch.from_numpy(sp).to(dev), torch.from_numpy(tp).to(dev)
    if is_bilinear_filter(w) and filters == real_in and s in (2, 8) and x.c == cp:
        if s == 2 and not bn_train:
            y = ops.upsample2x_relu_add(x, y=None, scale=scale, shift=shift, relu=relu)
        else:
            y = ops.upsample_raw_fwd(x, s)
            if scale is not None:
                _lib.check(_lib.lib().xv_bn_apply(y.xv(), ops._ptr(scale), ops._ptr(shift), int(relu), y.xv(), ops._stream()),
                           'xv_bn_apply')
            elif relu and not bn_train:
                ones = torch.ones(cp, dtype=torch.float32, device=dev)
                _lib.check(_lib.lib().xv_bn_apply(y.xv(), ops._ptr(ones), ops._ptr(torch.zeros_like(ones)), 1, y.xv(),
                                                  ops._stream()), 'xv_bn_apply')
    else:
        ckey = ('deconv', id(variables), wkey, x.c, str(dev))
        ent = _cache_get(ckey, variables)
        if ent is None:
            wp = np.zeros((k, k, cp, x.c), np.float32)
            wp[:, :, :filters, :real_in] = w
            ent = (ops.pack_conv_weights(torch.from_numpy(dense_deconv_as_conv3x3(wp, s)).to(dev)),
                   torch.zeros(s * s * cp, dtype=torch.float32, device=dev))
            _cache_put(ckey, variables, ent)
        y = ops.deconv_dense_fwd(x, ent[0], ent[1], s, cp, scale=scale, shift=shift, relu=relu and not bn_train)[0]
    if bn_train:
        st = _bn_training_state(variables, scope, name, cp, filters, dev)
        ops.bn_forward(y, st['gamma'], st['beta'], st['mm'], st['mv'], st['st'], y, relu=relu)
        _bn_write_back(variables, scope, name, st, filters)
    y.filters = filters
    return y
