"""PyTorch-CPU fp32 restatement of the reference FCN graph (TEST INFRASTRUCTURE, see
oracle/__init__.py).  NHWC numpy arrays in, NHWC numpy arrays out; weights in the
reference npz schema (HWIO conv kernels, [kh,kw,out,in] transposed-conv kernels).

`policy` mirrors where the MI355X path rounds to a storage dtype so that the oracle can
be fed *the same* rounded operands:
  'fp32'  every tensor fp32 (the reference's arithmetic)
  'bf16'  activations between layers and conv weights (all but conv1_1) rounded to
          bf16 (round-to-nearest-even), accumulation fp32, decoder head (upscore x8,
          score, softmax) fp32 from the bf16 `fused` features
  'fp8'   BASELINE config "fp8 MFMA conv path": as 'bf16', but the operands of conv2_2 .. conv5_3 and of the
          two 1x1 score convs are OCP e4m3fn with per-tensor power-of-two scales (`fp8_scales`: the output
          exponent of every map that is stored as fp8, 'w:<layer>' the weight exponents); conv1_1 (fp32),
          conv1_2 and conv2_1 (bf16 operands) are unchanged except that conv2_1 WRITES the first fp8 map
          (fp8_start='conv2_1': conv2_1 takes e4m3 operands too and conv1_2 writes the first fp8 map -- the plan of rounds
          2-4; fp8_deep: conv1_2 as well, conv1_1 writing the first fp8 map)
"""
import numpy as np
import torch
import torch.nn.functional as F

ENCODER_CONVS = [  # name, Cout  (simple_fcn.py:39-67)
    ('conv1_1', 64), ('conv1_2', 64), 'pool1',
    ('conv2_1', 128), ('conv2_2', 128), 'pool2',
    ('conv3_1', 256), ('conv3_2', 256), ('conv3_3', 256), 'pool3',
    ('conv4_1', 512), ('conv4_2', 512), ('conv4_3', 512), 'pool4',
    ('conv5_1', 512), ('conv5_2', 512), ('conv5_3', 512)]


def bilinear_1d(k):
    """1-D factor of bilinear_filter_initializer (custom_layers.py:8-25)."""
    factor = np.ceil(k / 2.0)
    center = (2 * factor - 1 - factor % 2) / (2.0 * factor)
    return np.array([1 - abs(x / factor - center) for x in range(k)], dtype=np.float64)


def bilinear_kernel(k, channels):
    """custom_layers.py:8-25: W[x,y,i,i] = w1[x]*w1[y], zero off the channel diagonal."""
    w1 = bilinear_1d(k)
    w = np.zeros((k, k, channels, channels))
    for i in range(channels):
        w[:, :, i, i] = np.outer(w1, w1)
    return w.astype(np.float32)


def glorot_uniform(rng, shape):
    """[TF1] default kernel initialiser of tf.layers.conv2d."""
    kh, kw, cin, cout = shape
    lim = np.sqrt(6.0 / (kh * kw * cin + kh * kw * cout))
    return rng.uniform(-lim, lim, size=shape).astype(np.float32)


def init_fcn_weights(prefix, in_channels, num_units, num_classes, seed=1, bias_scale=0.0):
    """Random-init weights in the reference npz schema (names: 'Synthia Rand Cityscapes
    Examples.ipynb':897-931).  bias_scale>0 gives non-zero biases (tests)."""
    rng = np.random.default_rng(seed)
    w = {}
    cin = in_channels
    for item in ENCODER_CONVS:
        if isinstance(item, str):
            continue
        name, cout = item
        w['%s/%s/kernel' % (prefix, name)] = glorot_uniform(rng, (3, 3, cin, cout))
        w['%s/%s/bias' % (prefix, name)] = (bias_scale * rng.standard_normal(cout)).astype(np.float32)
        cin = cout
    for name in ('score_conv4', 'score_conv5'):
        w['%s/%s/kernel' % (prefix, name)] = glorot_uniform(rng, (1, 1, 512, num_units))
        w['%s/%s/bias' % (prefix, name)] = (bias_scale * rng.standard_normal(num_units)).astype(np.float32)
    w['%s/upscore_conv5/kernel' % prefix] = bilinear_kernel(4, num_units)
    w['%s/upscore/kernel' % prefix] = bilinear_kernel(16, num_units)
    w['%s/score/kernel' % prefix] = glorot_uniform(rng, (1, 1, num_units, num_classes))
    w['%s/score/bias' % prefix] = (bias_scale * rng.standard_normal(num_classes)).astype(np.float32)
    return w


def _t(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def round_bf16(x):
    """Round-to-nearest-even to bf16, returned as fp32 (torch tensor or ndarray)."""
    if isinstance(x, np.ndarray):
        return _t(x.astype(np.float32)).bfloat16().float().numpy()
    return x.bfloat16().float()


FP8_MAX = 448.0            # largest finite OCP e4m3fn (1.75 * 2^8)
FP8_LAYERS = ('conv2_2', 'conv3_1', 'conv3_2', 'conv3_3', 'conv4_1', 'conv4_2', 'conv4_3', 'conv5_1', 'conv5_2',
              'conv5_3', 'score_conv4', 'score_conv5')       # convs whose operands are e4m3 in the 'fp8' policy


def round_e4m3(x, scale_exp=0):
    """Quantise to OCP e4m3fn with a per-tensor power-of-two scale and return the REAL values q * 2^scale_exp (fp32):
    q = round-to-nearest-even(x / 2^scale_exp) on the e4m3 grid (3 mantissa bits, normal exponents 2^-6 .. 2^8,
    subnormal step 2^-9), saturating at +-448.  Restated from the OCP 8-bit floating point specification; cross-checked
    against torch.float8_e4m3fn for in-range values in tests/test_oracle_golden.py."""
    tensor = not isinstance(x, np.ndarray)
    a = (x.numpy() if tensor else np.asarray(x)).astype(np.float64) * 2.0 ** (-scale_exp)
    mag = np.minimum(np.abs(a), FP8_MAX)
    with np.errstate(divide='ignore'):
        e = np.floor(np.log2(np.where(mag > 0, mag, 1.0)))
    step = 2.0 ** (np.maximum(e, -6.0) - 3.0)           # spacing of the grid around mag (subnormals: 2^-9)
    q = np.rint(mag / step) * step                      # np.rint rounds half to even
    out = (np.sign(a) * np.minimum(q, FP8_MAX) * 2.0 ** scale_exp).astype(np.float32)
    return _t(out) if tensor else out


def fp8_scale_exp(amax, margin_bits=0):
    """Smallest power-of-two exponent e with amax / 2^e <= 448 (+ margin_bits)."""
    amax = float(amax)
    return 0 if not amax > 0 else int(np.ceil(np.log2(amax / FP8_MAX))) + int(margin_bits)


def conv2d_same(x, w, b=None, relu=False):
    """tf.layers.conv2d(padding='same', strides 1) [+ relu]  (custom_layers.py:124-139).
    x: torch NCHW fp32, w: numpy HWIO."""
    k = w.shape[0]
    wt = _t(w).permute(3, 2, 0, 1).contiguous()
    y = F.conv2d(x, wt, None if b is None else _t(b), padding=(k - 1) // 2)
    return F.relu(y) if relu else y


def maxpool2(x):
    """max_pooling2d(2,2,'valid') (simple_fcn.py:41,44,48,58)."""
    return F.max_pool2d(x, 2, 2)


def deconv_same(x, w, stride, relu=False):
    """tf.layers.conv2d_transpose(padding='same', use_bias=False) with kernel layout
    [kh,kw,out,in] (custom_layers.py:71-121).  [TF1] out = stride*in, pad_before=(k-s)//2."""
    k = w.shape[0]
    wt = _t(w).permute(3, 2, 0, 1).contiguous()        # torch wants [in,out,kh,kw]
    y = F.conv_transpose2d(x, wt, stride=stride, padding=(k - stride) // 2)
    return F.relu(y) if relu else y


def fcn_forward(x_nhwc, weights, prefix, policy='fp32', keep=None, fp8_scales=None, fp8_deep=False, fp8_start=None):
    """fcn() = encoder + decoder (simple_fcn.py:137-170) with batchnorm=False, no dropout.
    Returns dict of NHWC fp32 numpy arrays.  keep: iterable of layer names to return
    (default: fused, upscore, score).  fp8_scales: {layer: output exponent, 'w:layer': weight exponent}
    for policy 'fp8' (missing weight exponents default to the smallest that keeps max|w| finite)."""
    assert policy in ('fp32', 'bf16', 'fp8')
    rnd = (lambda t: t) if policy == 'fp32' else round_bf16
    fp8_scales = fp8_scales or {}
    # 'fp8' policy (the product's fcn.fp8_plan): the convs from `fp8_start` on take e4m3 operands (and the two 1x1 score
    # convs, which write bf16), the conv in front of it and every conv from it on store their outputs as e4m3.  Default
    # 'conv2_2' (conv2_1 writes the first e4m3 map); 'conv2_1' = the plan of rounds 2-4; fp8_deep = 'conv1_2'.
    fp8_layers = FP8_LAYERS
    fp8_out = set()
    if policy == 'fp8':
        convs = [item[0] for item in ENCODER_CONVS if not isinstance(item, str)]
        i = convs.index(fp8_start or ('conv1_2' if fp8_deep else 'conv2_2'))
        assert i >= 1
        fp8_layers = tuple(convs[i:]) + ('score_conv4', 'score_conv5')
        fp8_out = set(convs[i - 1:])

    def rnd_out(t, name):
        return round_e4m3(t, fp8_scales[name]) if name in fp8_out else rnd(t)
    keep = set(keep) if keep is not None else {'fused', 'upscore', 'score'}
    out = {}

    def _bn(name):
        """Inference batch norm AFTER the conv and BEFORE the activation (custom_layers.py:126-137;
        [TF1] epsilon 1e-3), expressed as a per-channel scale/shift of the conv output."""
        g = weights.get('%s/%s/gamma' % (prefix, name))
        if g is None:
            return None
        s = g / np.sqrt(weights['%s/%s/moving_variance' % (prefix, name)] + 1e-3)
        return s.astype(np.float32), (weights['%s/%s/beta' % (prefix, name)] -
                                      weights['%s/%s/moving_mean' % (prefix, name)] * s).astype(np.float32)

    def W(name):
        w = weights['%s/%s/kernel' % (prefix, name)]
        bn = _bn(name)
        if bn is not None:          # y = s*(conv(x,W)+b)+t == conv(x, W*s) + (b*s+t): what the MI355X path folds
            w = w * bn[0]
        if policy == 'fp8' and name in fp8_layers:
            return round_e4m3(w, fp8_scales.get('w:' + name, fp8_scale_exp(np.abs(w).max())))
        if policy != 'fp32' and name != 'conv1_1':
            w = round_bf16(w)
        return w

    def B(name):
        b = weights['%s/%s/bias' % (prefix, name)]
        bn = _bn(name)
        return b if bn is None else (b * bn[0] + bn[1]).astype(np.float32)

    with torch.no_grad():
        h = _t(np.asarray(x_nhwc, np.float32)).permute(0, 3, 1, 2).contiguous()
        layers = {}
        for item in ENCODER_CONVS:
            if isinstance(item, str):
                h = maxpool2(h)
                layers[item] = h
            else:
                name = item[0]
                h = rnd_out(conv2d_same(h, W(name), B(name), relu=True), name)
                layers[name] = h
        score_conv4 = rnd(conv2d_same(layers['conv4_3'], W('score_conv4'), B('score_conv4'), relu=True))
        score_conv5 = rnd(conv2d_same(layers['conv5_3'], W('score_conv5'), B('score_conv5'), relu=True))
        # deconv kernels are exact in bf16 only for the 4x4 one; they are constants applied
        # in fp32 on the MI355X path, so they are never rounded here.
        def deconv_bn_relu(t, name, stride):
            """deconv -> [batch norm] -> relu (custom_layers.py:112-119)."""
            y = deconv_same(t, weights['%s/%s/kernel' % (prefix, name)], stride, relu=False)
            bn = _bn(name)
            if bn is not None:
                y = y * _t(bn[0]).view(1, -1, 1, 1) + _t(bn[1]).view(1, -1, 1, 1)
            return F.relu(y)

        up5 = deconv_bn_relu(score_conv5, 'upscore_conv5', 2)
        fused = rnd(score_conv4 + up5)                                   # tf.add_n (simple_fcn.py:85)
        layers.update(score_conv4=score_conv4, score_conv5=score_conv5, upscore_conv5=up5,
                      fused=fused)
        upscore = deconv_bn_relu(fused, 'upscore', 8)
        score = conv2d_same(upscore, W('score'), B('score'), relu=False)  # no activation
        layers.update(upscore=upscore, score=score)
        for k in keep:
            out[k] = layers[k].permute(0, 2, 3, 1).contiguous().numpy()
    return out


def softmax(score):
    """tf.nn.softmax over the last axis, fp32: exp(x-max)/sum(exp(x-max))."""
    s = np.asarray(score, np.float32)
    e = np.exp(s - s.max(axis=-1, keepdims=True), dtype=np.float32)
    return (e / e.sum(axis=-1, keepdims=True, dtype=np.float32)).astype(np.float32)


def log_softmax(score):
    s = np.asarray(score, np.float32)
    z = s - s.max(axis=-1, keepdims=True)
    return (z - np.log(np.exp(z, dtype=np.float32).sum(axis=-1, keepdims=True, dtype=np.float32))).astype(np.float32)


def argmax_last(x):
    """tf.argmax(.,3): int64, lowest index among ties."""
    return np.argmax(x, axis=-1).astype(np.int64)


def cross_entropy(log_p, labels_int, num_classes):
    """utils.py:43-53 with one_hot(label) (base_model.py:198-201): labels outside [0,C)
    give an all-zero row and drop out.  Returns (loss, dloss/dscore) in fp64/fp32."""
    lab = np.asarray(labels_int)
    valid = (lab >= 0) & (lab < num_classes)
    onehot = np.zeros(log_p.shape, np.float32)
    idx = np.nonzero(valid)
    onehot[idx + (lab[idx],)] = 1.0
    denom = 1e-20 + float(onehot.sum())
    loss = -float((onehot.astype(np.float64) * log_p.astype(np.float64)).sum()) / denom
    # d loss / d score = (softmax * sum(onehot,-1) - onehot) / denom
    p = np.exp(log_p.astype(np.float64))
    grad = (p * onehot.sum(-1, keepdims=True) - onehot) / denom
    return loss, grad.astype(np.float32)


# ---------------------------------------------------------------------------------------
# independent pure-numpy loop twin (small cases only): restates the TF op definitions
# directly, sharing no code with the torch path above.
# ---------------------------------------------------------------------------------------

def naive_conv2d_same(x, w, b=None, relu=False):
    n, h, wd, cin = x.shape
    k = w.shape[0]
    p = (k - 1) // 2
    cout = w.shape[3]
    xp = np.zeros((n, h + k - 1, wd + k - 1, cin), np.float64)
    xp[:, p:p + h, p:p + wd] = x
    y = np.zeros((n, h, wd, cout), np.float64)
    for dy in range(k):
        for dx in range(k):
            y += xp[:, dy:dy + h, dx:dx + wd, :] @ w[dy, dx].astype(np.float64)
    if b is not None:
        y += b
    return np.maximum(y, 0) if relu else y


def naive_maxpool2(x):
    n, h, w, c = x.shape
    return x.reshape(n, h // 2, 2, w // 2, 2, c).max(axis=(2, 4))


def naive_deconv_same(x, w, stride, relu=False):
    """out[o] += in[i] * W[p, :, out_c, in_c], o = i*s + p - (k-s)//2   (SURVEY App. B)."""
    n, h, wd, cin = x.shape
    k = w.shape[0]
    cout = w.shape[2]
    pb = (k - stride) // 2
    y = np.zeros((n, h * stride, wd * stride, cout), np.float64)
    for iy in range(h):
        for ix in range(wd):
            for py in range(k):
                oy = iy * stride + py - pb
                if oy < 0 or oy >= h * stride:
                    continue
                for px in range(k):
                    ox = ix * stride + px - pb
                    if ox < 0 or ox >= wd * stride:
                        continue
                    y[:, oy, ox, :] += x[:, iy, ix, :].astype(np.float64) @ w[py, px].astype(np.float64).T
    return np.maximum(y, 0) if relu else y


def depthwise_bilinear_up(x, stride, relu=False):
    """What the dense deconv collapses to when the kernel is the (diagonal) bilinear
    constant: out[o] = sum_{i,p: i*s+p-pb=o} x[i]*w1[p] per axis, zero outside."""
    k = 2 * stride
    w1 = bilinear_1d(k)
    pb = (k - stride) // 2
    n, h, wd, c = x.shape

    def up_axis(a, axis):
        a = np.moveaxis(a, axis, 0)
        L = a.shape[0]
        out = np.zeros((L * stride,) + a.shape[1:], np.float64)
        for i in range(L):
            for p in range(k):
                o = i * stride + p - pb
                if 0 <= o < L * stride:
                    out[o] += a[i] * w1[p]
        return np.moveaxis(out, 0, axis)

    y = up_axis(up_axis(x.astype(np.float64), 1), 2)
    return np.maximum(y, 0) if relu else y


# ---------------------------------------------------------------------------------------
# training step oracle: loss of SimpleFCN._build_graph (simple_fcn.py:200-214) and its
# gradients by PyTorch-CPU autograd over the same functional graph (fp32).
# ---------------------------------------------------------------------------------------

class _RoundBf16STE(torch.autograd.Function):
    """bf16 rounding in the forward pass, identity in the backward pass."""

    @staticmethod
    def forward(ctx, t):
        return t.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return g


def fcn_loss_and_grads(x_nhwc, labels, weights, prefix, num_classes, policy='fp32', batch_norm=False, layers_out=None):
    """Returns (loss, {variable name: gradient ndarray}) for the trainable variables
    (kernels + biases of the 13 convs, the two score convs and `score`; the deconv
    constants are not trainable, simple_fcn.py:80-83,117-119).  policy='bf16' rounds the
    forward activations / conv weights where the MI355X path does (straight-through), so that
    relu masks and max-pool routing are those of the bf16 forward pass.
    batch_norm=True: the training graph of `batch_normalization: true` (custom_layers.py:112-119,124-139):
    tf.layers.batch_normalization(training=True) between every conv / deconv and its activation -- statistics
    over (N, H, W), biased variance, eps 1e-3 -- with trainable gamma / beta ([TF1]); the third return value is
    then {layer: (batch mean, unbiased batch variance)} (what the moving averages are fed)."""
    params, stats = {}, {}
    rnd = (lambda t: t) if policy == 'fp32' else _RoundBf16STE.apply

    def P(name):
        t = _t(weights[name]).clone().requires_grad_(True)
        params[name] = t
        return t

    def bn(y, layer):
        if not batch_norm:
            return y
        g, b = P('%s/%s/gamma' % (prefix, layer)), P('%s/%s/beta' % (prefix, layer))
        mean = y.mean(dim=(0, 2, 3), keepdim=True)
        var = ((y - mean) ** 2).mean(dim=(0, 2, 3), keepdim=True)
        m = y.shape[0] * y.shape[2] * y.shape[3]
        stats[layer] = (mean.detach().numpy().ravel(), (var.detach().numpy().ravel() * m / max(m - 1, 1)))
        return (y - mean) / torch.sqrt(var + 1e-3) * g.view(1, -1, 1, 1) + b.view(1, -1, 1, 1)

    def conv(h, layer, relu):
        w = P('%s/%s/kernel' % (prefix, layer))
        b = P('%s/%s/bias' % (prefix, layer))
        k = w.shape[0]
        wq = w if layer in ('conv1_1', 'score') else rnd(w)
        y = F.conv2d(h, wq.permute(3, 2, 0, 1), b, padding=(k - 1) // 2)
        if batch_norm:
            y = bn(y if layer == 'score' else rnd(y), layer)      # the pre-normalisation map is stored (bf16) too
        y = F.relu(y) if relu else y
        return y if layer == 'score' else rnd(y)

    def deconv(h, layer, stride):
        w = _t(weights['%s/%s/kernel' % (prefix, layer)])
        k = w.shape[0]
        y = F.conv_transpose2d(h, w.permute(3, 2, 0, 1).contiguous(), stride=stride, padding=(k - stride) // 2)
        if batch_norm:
            y = rnd(bn(rnd(y), layer))
        return F.relu(y)

    h = _t(np.asarray(x_nhwc, np.float32)).permute(0, 3, 1, 2).contiguous()
    layers = {}
    for item in ENCODER_CONVS:
        if isinstance(item, str):
            h = F.max_pool2d(h, 2, 2)
        else:
            h = conv(h, item[0], True)
        layers[item if isinstance(item, str) else item[0]] = h
    s4 = conv(layers['conv4_3'], 'score_conv4', True)
    s5 = conv(layers['conv5_3'], 'score_conv5', True)
    fused = rnd(s4 + deconv(s5, 'upscore_conv5', 2))
    up = deconv(fused, 'upscore', 8)
    score = conv(up, 'score', False).permute(0, 2, 3, 1)
    logp = F.log_softmax(score, dim=-1)
    lab = _t(np.asarray(labels).astype(np.int64))
    valid = (lab >= 0) & (lab < num_classes)
    onehot = F.one_hot(lab.clamp(0, num_classes - 1), num_classes).float() * valid[..., None].float()
    loss = -(onehot * logp).sum() / (1e-20 + onehot.sum())          # utils.py:43-53
    loss.backward()
    if layers_out is not None:       # post-activation maps (NHWC) for debugging / tests
        layers.update(score_conv4=s4, score_conv5=s5, fused=fused, upscore=up)
        for k, v in layers.items():
            layers_out[k] = v.detach().permute(0, 2, 3, 1).numpy()
        layers_out['score'] = score.detach().numpy()
    grads = {k: v.grad.numpy() for k, v in params.items()}
    if batch_norm:
        return float(loss.detach()), grads, stats
    return float(loss.detach()), grads


# ---- fusion_fcn: the joint two-stream baseline (xview/models/fusion_fcn.py:11-40, vgg16.py:7-51) --------------
def init_fusion_fcn_weights(prefixes, num_channels, num_units, num_classes, seed=1, bias_scale=0.0, bn=True):
    """Random-init weights under the variable names the reference graph creates: trunks `{prefix}_convX_Y`
    (vgg16.py:18-37), `fused_score_conv4/5`, `fused_upscore_conv5` (fusion_fcn.py:30-36) and the decoder's
    `fused/upscore`, `fused/score` (simple_fcn.py:121-133), which -- decoder() being called without `batchnorm` --
    carry batch norm variables (bn=True: random moving statistics)."""
    rng = np.random.default_rng(seed)
    w = {}
    for m, prefix in prefixes.items():
        cin = num_channels[m]
        for item in ENCODER_CONVS:
            if isinstance(item, str):
                continue
            name, cout = item
            w['%s_%s/kernel' % (prefix, name)] = glorot_uniform(rng, (3, 3, cin, cout))
            w['%s_%s/bias' % (prefix, name)] = (bias_scale * rng.standard_normal(cout)).astype(np.float32)
            cin = cout
    e = len(prefixes)
    for name in ('fused_score_conv4', 'fused_score_conv5'):
        w[name + '/kernel'] = glorot_uniform(rng, (1, 1, 512 * e, num_units))
        w[name + '/bias'] = (bias_scale * rng.standard_normal(num_units)).astype(np.float32)
    w['fused_upscore_conv5/kernel'] = bilinear_kernel(4, num_units)
    w['fused/upscore/kernel'] = bilinear_kernel(16, num_units)
    w['fused/score/kernel'] = glorot_uniform(rng, (1, 1, num_units, num_classes))
    w['fused/score/bias'] = (bias_scale * rng.standard_normal(num_classes)).astype(np.float32)
    if bn:
        for layer, c in (('fused/upscore', num_units), ('fused/score', num_classes)):
            w[layer + '/gamma'] = rng.uniform(0.7, 1.3, c).astype(np.float32)
            w[layer + '/beta'] = (0.1 * rng.standard_normal(c)).astype(np.float32)
            w[layer + '/moving_mean'] = (0.1 * rng.standard_normal(c)).astype(np.float32)
            w[layer + '/moving_variance'] = rng.uniform(0.6, 1.4, c).astype(np.float32)
    return w


def fusion_fcn_forward(inputs, weights, prefixes, policy='fp32', keep=None):
    """fusion_fcn() at inference (fusion_fcn.py:11-40): one VGG16 trunk per modality (no batch norm), channel
    concat of the conv4_3 / conv5_3 maps, 1x1 `fused_score_conv4/5` + relu, x2 bilinear deconv + relu, add, then
    decoder(features, 'fused') WITH its default batch norm: upscore = relu(BN(deconv x8)), score = BN(conv1x1).
    inputs: {modality: NHWC fp32}.  Returns dict of NHWC fp32 arrays (default keys: features, score)."""
    assert policy in ('fp32', 'bf16')
    rnd = (lambda t: t) if policy == 'fp32' else round_bf16
    keep = set(keep) if keep is not None else {'features', 'score'}

    def bn(layer):
        g = weights.get(layer + '/gamma')
        if g is None:
            return None
        s = g / np.sqrt(weights[layer + '/moving_variance'] + 1e-3)
        return s.astype(np.float32), (weights[layer + '/beta'] - weights[layer + '/moving_mean'] * s).astype(np.float32)

    def W(layer, first=False):
        w = weights[layer + '/kernel']
        return w if (policy == 'fp32' or first) else round_bf16(w)

    with torch.no_grad():
        layers = {}
        c4, c5 = [], []
        for m, prefix in prefixes.items():
            h = _t(np.asarray(inputs[m], np.float32)).permute(0, 3, 1, 2).contiguous()
            for item in ENCODER_CONVS:
                if isinstance(item, str):
                    h = maxpool2(h)
                else:
                    layer = '%s_%s' % (prefix, item[0])
                    h = rnd(conv2d_same(h, W(layer, item[0] == 'conv1_1'), weights[layer + '/bias'], relu=True))
                    if item[0] == 'conv4_3':
                        c4.append(h)
            c5.append(h)
        concat4, concat5 = torch.cat(c4, dim=1), torch.cat(c5, dim=1)          # tf.concat(axis=3) in NHWC
        s4 = rnd(conv2d_same(concat4, W('fused_score_conv4'), weights['fused_score_conv4/bias'], relu=True))
        s5 = rnd(conv2d_same(concat5, W('fused_score_conv5'), weights['fused_score_conv5/bias'], relu=True))
        up5 = deconv_same(s5, weights['fused_upscore_conv5/kernel'], 2, relu=True)
        features = rnd(s4 + up5)
        y = deconv_same(features, weights['fused/upscore/kernel'], 8, relu=False)
        b = bn('fused/upscore')
        if b is not None:
            y = y * _t(b[0]).view(1, -1, 1, 1) + _t(b[1]).view(1, -1, 1, 1)
        upscore = F.relu(y)
        score = conv2d_same(upscore, weights['fused/score/kernel'], weights['fused/score/bias'], relu=False)
        b = bn('fused/score')
        if b is not None:
            score = score * _t(b[0]).view(1, -1, 1, 1) + _t(b[1]).view(1, -1, 1, 1)
        layers.update(concat_conv4=concat4, concat_conv5=concat5, score_conv4=s4, score_conv5=s5, upscore_conv5=up5,
                      features=features, upscore=upscore, score=score)
        return {k: layers[k].permute(0, 2, 3, 1).contiguous().numpy() for k in keep}


def fusion_fcn_loss_and_grads(inputs, labels, weights, prefixes, num_classes, policy='fp32'):
    """Training graph of the joint model (fusion_fcn.py:11-40 with is_training=True, loss of FusionFCN._build_graph
    :87-92 = models/utils.py:43-53): trunks and the fused 1x1 convs without batch norm, decoder() with its default
    batch norm in training mode on `fused/upscore` (between the x8 deconv and its relu) and `fused/score`.
    Returns (loss, {variable: gradient}, {bn layer: (batch mean, unbiased batch variance)}).  policy as in
    fcn_loss_and_grads."""
    params, stats = {}, {}
    rnd = (lambda t: t) if policy == 'fp32' else _RoundBf16STE.apply

    def P(name):
        t = _t(weights[name]).clone().requires_grad_(True)
        params[name] = t
        return t

    def bn(y, layer):
        g, b = P(layer + '/gamma'), P(layer + '/beta')
        mean = y.mean(dim=(0, 2, 3), keepdim=True)
        var = ((y - mean) ** 2).mean(dim=(0, 2, 3), keepdim=True)
        m = y.shape[0] * y.shape[2] * y.shape[3]
        stats[layer] = (mean.detach().numpy().ravel(), var.detach().numpy().ravel() * m / max(m - 1, 1))
        return (y - mean) / torch.sqrt(var + 1e-3) * g.view(1, -1, 1, 1) + b.view(1, -1, 1, 1)

    def conv(h, layer, first=False):
        w, b = P(layer + '/kernel'), P(layer + '/bias')
        k = w.shape[0]
        y = F.conv2d(h, (w if first else rnd(w)).permute(3, 2, 0, 1), b, padding=(k - 1) // 2)
        return rnd(F.relu(y))

    def deconv(h, layer, stride):
        w = _t(weights[layer + '/kernel'])
        k = w.shape[0]
        return F.conv_transpose2d(h, w.permute(3, 2, 0, 1).contiguous(), stride=stride, padding=(k - stride) // 2)

    c4, c5 = [], []
    for m, prefix in prefixes.items():
        h = _t(np.asarray(inputs[m], np.float32)).permute(0, 3, 1, 2).contiguous()
        for item in ENCODER_CONVS:
            if isinstance(item, str):
                h = F.max_pool2d(h, 2, 2)
            else:
                h = conv(h, '%s_%s' % (prefix, item[0]), first=item[0] == 'conv1_1')
                if item[0] == 'conv4_3':
                    c4.append(h)
        c5.append(h)
    s4 = conv(torch.cat(c4, dim=1), 'fused_score_conv4')
    s5 = conv(torch.cat(c5, dim=1), 'fused_score_conv5')
    features = rnd(s4 + F.relu(deconv(s5, 'fused_upscore_conv5', 2)))
    up = rnd(F.relu(bn(rnd(deconv(features, 'fused/upscore', 8)), 'fused/upscore')))
    w, b = P('fused/score/kernel'), P('fused/score/bias')
    score = bn(F.conv2d(up, w.permute(3, 2, 0, 1), b), 'fused/score').permute(0, 2, 3, 1)
    logp = F.log_softmax(score, dim=-1)
    lab = _t(np.asarray(labels).astype(np.int64))
    valid = (lab >= 0) & (lab < num_classes)
    onehot = F.one_hot(lab.clamp(0, num_classes - 1), num_classes).float() * valid[..., None].float()
    loss = -(onehot * logp).sum() / (1e-20 + onehot.sum())
    loss.backward()
    return float(loss.detach()), {k: v.grad.numpy() for k, v in params.items()}, stats
