"""CPU restatement of the reference's AdapNet expert (inference graph) -- TEST INFRASTRUCTURE ONLY.

Follows /root/reference xview/models/adapnet.py op for op (block_a :12-51, block_b :54-100, adapnet
:103-173, Adapnet._build_graph test branch :212-219) with the [TF1] semantics the reference inherits:
`padding='same'` with stride 2 pads (k-2)//2 before and the rest after on even inputs (7x7: 2 / 3;
1x1: none), dilated 3x3 'same' pads by the dilation rate, batch normalisation (epsilon 1e-3) sits
between every convolution and its activation (custom_layers.py:124-139), both deconvolutions are
constant bilinear kernels of shape [k,k,filters,in] whose only non-zero entries are [.,.,i,i] for
i < filters (custom_layers.py:8-25,71-121) followed by batch normalisation and no activation.

Parity is pinned the same way as fcn_oracle: TensorFlow is absent, so this restatement is checked
against the reference's published building blocks only through shared pieces (bilinear kernels,
batch-norm folding, softmax/argmax) -- "golden-by-oracle" for the full graph.
Only tests/ may import this module.
"""
import numpy as np
import torch
import torch.nn.functional as F

from .fcn_oracle import _t, bilinear_1d, glorot_uniform, round_bf16

# (name, kind, arguments): a = block_a(intermediate, filters, stride, shortcut_conv),
# b = block_b(filters_1, filters_2, filters_3, dilation1, dilation2, shortcut_conv)   (adapnet.py:130-155)
BLOCKS = [
    ('block_layer_1', 'a', (64, 256, 1, True)), ('block_layer_2', 'a', (64, 256, 1, False)),
    ('block_layer_3', 'a', (64, 256, 1, False)), ('block_layer_4', 'a', (128, 512, 2, True)),
    ('block_layer_5', 'a', (128, 512, 1, False)), ('block_layer_6', 'a', (128, 512, 1, False)),
    ('block_layer_7', 'b', (128, 64, 512, 1, 2, False)),
    ('block_layer_8', 'a', (256, 1024, 2, True)), ('block_layer_9', 'a', (256, 1024, 1, False)),
    ('block_layer_10', 'b', (256, 256, 1024, 1, 2, False)), ('block_layer_11', 'b', (256, 256, 1024, 1, 4, False)),
    ('block_layer_12', 'b', (256, 256, 1024, 1, 8, False)), ('block_layer_13', 'b', (256, 256, 1024, 1, 16, False)),
    ('block_layer_14', 'b', (512, 512, 2048, 2, 4, True)), ('block_layer_15', 'b', (512, 512, 2048, 2, 8, False)),
    ('block_layer_16', 'b', (512, 512, 2048, 2, 16, False)),
]


def conv_layers(in_channels, num_units, blocks=None):
    """Every convolution of the graph as (variable scope, k, cin, cout, has_bias), in graph order."""
    layers = [('block_0_1', 3, in_channels, 64, True), ('block_0_2', 7, 64, 64, True)]
    cin = 64
    for name, kind, args in (blocks or BLOCKS):
        if kind == 'a':
            mid, cout, _, shortcut = args
            layers += [(name + '/stage_1', 1, cin, mid, False), (name + '/stage_2', 3, mid, mid, False),
                       (name + '/stage_3', 1, mid, cout, False)]
        else:
            f1, f2, cout, _, _, shortcut = args
            layers += [(name + '/stage_1', 1, cin, f1, False), (name + '/stage_2_1', 3, f1, f2 // 2, False),
                       (name + '/stage_2_2', 3, f1, f2 // 2, False), (name + '/stage_3', 1, f2, cout, False)]
        if shortcut:
            layers.append((name + '/shortcut', 1, cin, cout, False))
        if name == 'block_layer_7':
            layers.append(('shortcut', 1, cout, num_units, True))
        cin = cout
    layers.append(('first_deconvolution_conv', 1, cin, cin, True))          # 2048 -> 2048 in the reference graph
    return layers


def rect_bilinear_kernel(k, filters, in_channels):
    """custom_layers.py:8-25 for the [k,k,filters,in] kernels AdapNet asks for."""
    w1 = bilinear_1d(k)
    w = np.zeros((k, k, filters, in_channels))
    for i in range(filters):
        w[:, :, i, i] = np.outer(w1, w1)
    return w.astype(np.float32)


def init_adapnet_weights(prefix, in_channels, num_units, num_classes, seed=1, gain=1.0, blocks=None):
    """Random weights in the reference's npz schema: '<prefix>/<scope>/{kernel,bias,gamma,beta,moving_mean,
    moving_variance}' (conv and its batch norm share the scope name, custom_layers.py:131-135)."""
    rng = np.random.default_rng(seed)
    w = {}

    def bn(scope, c):
        w['%s/%s/gamma' % (prefix, scope)] = rng.uniform(0.6, 1.4, c).astype(np.float32)
        w['%s/%s/beta' % (prefix, scope)] = (0.1 * rng.standard_normal(c)).astype(np.float32)
        w['%s/%s/moving_mean' % (prefix, scope)] = (0.1 * rng.standard_normal(c)).astype(np.float32)
        w['%s/%s/moving_variance' % (prefix, scope)] = rng.uniform(0.5, 1.5, c).astype(np.float32)

    layers = conv_layers(in_channels, num_units, blocks)
    for scope, k, cin, cout, has_bias in layers:
        w['%s/%s/kernel' % (prefix, scope)] = glorot_uniform(rng, (k, k, cin, cout)) * gain
        if has_bias:
            w['%s/%s/bias' % (prefix, scope)] = (0.02 * rng.standard_normal(cout)).astype(np.float32)
        bn(scope, cout)
    w['%s/first_deconvolution_upconv/kernel' % prefix] = rect_bilinear_kernel(4, num_units, layers[-1][3])
    bn('first_deconvolution_upconv', num_units)
    w['%s/second_deconvolution_upconv/kernel' % prefix] = rect_bilinear_kernel(16, num_classes, num_units)
    bn('second_deconvolution_upconv', num_classes)
    return w


def adapnet_forward(x_nhwc, weights, prefix, policy='fp32', keep=None, blocks=None):
    """adapnet(..., is_training=False) (adapnet.py:103-173); dict of NHWC fp32 arrays for the layers in `keep`
    (default: merge, score).  policy 'bf16': folded conv weights (all but block_0_1) and the activations
    between layers rounded to bf16, accumulation and the two deconvolution stages fp32 -- the MI355X path's
    storage policy."""
    assert policy in ('fp32', 'bf16')
    rnd = (lambda t: t) if policy == 'fp32' else round_bf16
    keep = set(keep) if keep is not None else {'merge', 'score'}

    def affine(scope):
        s = weights['%s/%s/gamma' % (prefix, scope)] / np.sqrt(weights['%s/%s/moving_variance' % (prefix, scope)] + 1e-3)
        t = weights['%s/%s/beta' % (prefix, scope)] - weights['%s/%s/moving_mean' % (prefix, scope)] * s
        return s.astype(np.float32), t.astype(np.float32)

    def conv(h, scope, stride=1, dilation=1, relu=True):
        """conv2d -> batch norm -> activation (custom_layers.py:124-139), the batch norm folded into kernel and bias."""
        w = weights['%s/%s/kernel' % (prefix, scope)]
        s, t = affine(scope)
        b = weights.get('%s/%s/bias' % (prefix, scope))
        wf = w * s
        bf = t if b is None else b * s + t
        if policy == 'bf16' and scope != 'block_0_1':
            wf = round_bf16(wf)
        k = w.shape[0]
        total = max((k - 1) * dilation + 1 - stride, 0)              # [TF1] 'same' on inputs divisible by the stride
        before = total // 2
        h = F.pad(h, (before, total - before, before, total - before))
        y = F.conv2d(h, _t(wf).permute(3, 2, 0, 1).contiguous(), _t(bf), stride=stride, dilation=dilation)
        return F.relu(y) if relu else y

    def deconv_bn(h, scope, stride):
        w = weights['%s/%s/kernel' % (prefix, scope)]                 # [k,k,filters,in]
        k = w.shape[0]
        # a kernel that is still the bilinear constant is applied in fp32 (depthwise interpolation on the MI355X path);
        # a trained (dense) one runs on the bf16 MFMA conv: kernel and conv output rounded to bf16 under policy 'bf16'
        dense = not np.allclose(w, rect_bilinear_kernel(k, w.shape[2], w.shape[3]), atol=1e-6)
        if policy == 'bf16' and dense:
            w = round_bf16(w)
        y = F.conv_transpose2d(h, _t(w).permute(3, 2, 0, 1).contiguous(), stride=stride, padding=(k - stride) // 2)
        if dense:
            y = rnd(y)
        s, t = affine(scope)
        return y * _t(s).view(1, -1, 1, 1) + _t(t).view(1, -1, 1, 1)

    out = {}
    with torch.no_grad():
        h = _t(np.asarray(x_nhwc, np.float32)).permute(0, 3, 1, 2).contiguous()
        layers = {}
        h = layers['block_0_1'] = rnd(conv(h, 'block_0_1'))
        h = layers['block_0_2'] = rnd(conv(h, 'block_0_2', stride=2))
        h = layers['block_0_pool'] = F.max_pool2d(h, 2, 2)
        for index, (name, kind, args) in enumerate(blocks or BLOCKS, start=1):
            if kind == 'a':
                _, _, stride, shortcut_conv = args
                s1 = rnd(conv(h, name + '/stage_1', stride=stride))
                s2 = rnd(conv(s1, name + '/stage_2'))
            else:
                _, _, _, d1, d2, shortcut_conv = args
                stride = 1
                s1 = rnd(conv(h, name + '/stage_1'))
                s2 = torch.cat([rnd(conv(s1, name + '/stage_2_1', dilation=d1)),
                                rnd(conv(s1, name + '/stage_2_2', dilation=d2))], dim=1)
            s3 = conv(s2, name + '/stage_3')                          # relu INSIDE the branch, before the add
            short = rnd(conv(h, name + '/shortcut', stride=stride)) if shortcut_conv else h
            h = layers['block_%d' % index] = rnd(F.relu(s3 + short))
            if name == 'block_layer_7':
                layers['shortcut'] = rnd(conv(h, 'shortcut', relu=False))
        d = rnd(conv(h, 'first_deconvolution_conv'))
        layers['deconv_1'] = deconv_bn(d, 'first_deconvolution_upconv', 2)
        layers['merge'] = rnd(layers['deconv_1'] + layers['shortcut'])
        layers['score'] = deconv_bn(layers['merge'], 'second_deconvolution_upconv', 8)
        for k in keep:
            out[k] = layers[k].permute(0, 2, 3, 1).contiguous().numpy()
    return out


def adapnet_loss_and_grads(x_nhwc, labels, weights, prefix, num_classes, policy='fp32', units=None, blocks=None):
    """Training graph of the AdapNet expert: adapnet(..., is_training=True) (adapnet.py:103-173) with
    tf.layers.batch_normalization(training=True) after every conv / deconv -- statistics over (N, H, W), biased
    variance, eps 1e-3, trainable gamma / beta -- and the loss of Adapnet._build_graph (adapnet.py:196-203):
    cross_entropy (models/utils.py:43-53, already a mean over the labelled pixels) divided once more by the number
    of labelled pixels.  Returns (loss, {variable: gradient}, {bn scope: (batch mean, unbiased batch variance)}).
    The two deconv kernels are parameters too (the reference trains them).  `units` (default: all channels) restricts
    `first_deconvolution_conv` to its first `units` output channels -- only meaningful while the x2 deconv kernel is
    still the bilinear constant AND its own gradient is not looked at (the kernel's gradient is non-zero for every
    input channel).
    policy 'bf16': straight-through rounding where the MI355X path stores bf16 (see fcn_oracle.fcn_loss_and_grads)."""
    from .fcn_oracle import _RoundBf16STE
    params, stats = {}, {}
    rnd = (lambda t: t) if policy == 'fp32' else _RoundBf16STE.apply

    def P(name, sl=None):
        if name not in params:
            params[name] = _t(weights[name]).clone().requires_grad_(True)
        return params[name] if sl is None else params[name][..., :sl]

    def bn(y, scope, sl=None):
        g, b = P('%s/%s/gamma' % (prefix, scope), sl), P('%s/%s/beta' % (prefix, scope), sl)
        mean = y.mean(dim=(0, 2, 3), keepdim=True)
        var = ((y - mean) ** 2).mean(dim=(0, 2, 3), keepdim=True)
        m = y.shape[0] * y.shape[2] * y.shape[3]
        stats[scope] = (mean.detach().numpy().ravel(), var.detach().numpy().ravel() * m / max(m - 1, 1))
        return (y - mean) / torch.sqrt(var + 1e-3) * g.view(1, -1, 1, 1) + b.view(1, -1, 1, 1)

    def conv(h, scope, stride=1, dilation=1, relu=True, sl=None):
        w = P('%s/%s/kernel' % (prefix, scope), sl)
        has_bias = '%s/%s/bias' % (prefix, scope) in weights
        b = P('%s/%s/bias' % (prefix, scope), sl) if has_bias else None
        k = w.shape[0]
        total = max((k - 1) * dilation + 1 - stride, 0)
        before = total // 2
        h = F.pad(h, (before, total - before, before, total - before))
        wq = w if scope == 'block_0_1' else rnd(w)
        y = bn(rnd(F.conv2d(h, wq.permute(3, 2, 0, 1), b, stride=stride, dilation=dilation)), scope, sl)
        return rnd(F.relu(y) if relu else y)

    def deconv_bn(h, scope, stride):
        # the reference TRAINS both transposed-conv kernels (adapnet.py:155-163: deconv2d without trainable=False,
        # custom_layers.py:71-121, use_bias=False); policy 'bf16': kernel and conv output rounded like every MFMA conv
        w = P('%s/%s/kernel' % (prefix, scope))
        k = w.shape[0]
        return rnd(F.conv_transpose2d(h, rnd(w).permute(3, 2, 0, 1).contiguous(), stride=stride, padding=(k - stride) // 2))

    h = _t(np.asarray(x_nhwc, np.float32)).permute(0, 3, 1, 2).contiguous()
    h = conv(h, 'block_0_1')
    h = conv(h, 'block_0_2', stride=2)
    h = F.max_pool2d(h, 2, 2)
    shortcut = None
    for name, kind, args in (blocks or BLOCKS):
        if kind == 'a':
            _, _, stride, shortcut_conv = args
            s1 = conv(h, name + '/stage_1', stride=stride)
            s2 = conv(s1, name + '/stage_2')
        else:
            _, _, _, d1, d2, shortcut_conv = args
            stride = 1
            s1 = conv(h, name + '/stage_1')
            s2 = torch.cat([conv(s1, name + '/stage_2_1', dilation=d1), conv(s1, name + '/stage_2_2', dilation=d2)], dim=1)
        s3 = conv(s2, name + '/stage_3')
        short = conv(h, name + '/shortcut', stride=stride) if shortcut_conv else h
        h = rnd(F.relu(s3 + short))
        if name == 'block_layer_7':
            shortcut = conv(h, 'shortcut', relu=False)
    u = weights['%s/first_deconvolution_upconv/kernel' % prefix].shape[2]
    d = conv(h, 'first_deconvolution_conv', sl=units)
    width = weights['%s/first_deconvolution_conv/kernel' % prefix].shape[3]
    if units is not None and units < width:
        d = F.pad(d, (0, 0, 0, 0, 0, width - d.shape[1]))              # the unused channels: any value, zero weight
    deconv_1 = rnd(bn(rnd(deconv_bn(d, 'first_deconvolution_upconv', 2)), 'first_deconvolution_upconv'))
    merge = rnd(deconv_1 + shortcut)
    score = bn(deconv_bn(merge, 'second_deconvolution_upconv', 8), 'second_deconvolution_upconv').permute(0, 2, 3, 1)
    assert deconv_1.shape[1] == u
    logp = F.log_softmax(score, dim=-1)
    lab = _t(np.asarray(labels).astype(np.int64))
    valid = (lab >= 0) & (lab < num_classes)
    onehot = F.one_hot(lab.clamp(0, num_classes - 1), num_classes).float() * valid[..., None].float()
    ce = -(onehot * logp).sum() / (1e-20 + onehot.sum())
    loss = ce / onehot.sum()                                                # adapnet.py:202-203
    loss.backward()
    grads = {k: (v.grad.numpy() if v.grad is not None else np.zeros(v.shape, np.float32)) for k, v in params.items()}
    return float(loss.detach()), grads, stats
