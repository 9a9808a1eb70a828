"""The FCN expert on MI355X: host-side mirror of `encoder` / `decoder` / `fcn`
(xview/models/simple_fcn.py:10-170) driving the HIP kernels through the C ABI.

Data layout in HBM: bf16 padded-NHWC activations (see ops.Act), packed bf16 conv weights,
fp32 first-layer / score weights.  Per image and stream the encoder runs 13 convs (12 on MFMA;
pool1..pool4 fused into the epilogues of conv1_2/2_2/3_3/4_3), two 1x1 score convs, the x2
bilinear + add, and ONE fused decoder-head kernel (x8 bilinear + relu + score + softmax + argmax).
"""
import os

import numpy as np
import torch

from . import ops
from .custom_layers import bilinear_filter, dense_deconv_as_conv3x3, is_bilinear_filter

ENCODER = [  # (name, cout, pool_after)   simple_fcn.py:39-67
    ('conv1_1', 64, None), ('conv1_2', 64, 'pool1'),
    ('conv2_1', 128, None), ('conv2_2', 128, 'pool2'),
    ('conv3_1', 256, None), ('conv3_2', 256, None), ('conv3_3', 256, 'pool3'),
    ('conv4_1', 512, None), ('conv4_2', 512, None), ('conv4_3', 512, 'pool4'),
    ('conv5_1', 512, None), ('conv5_2', 512, None), ('conv5_3', 512, None)]
BN_EPS = 1e-3  # [TF1] tf.layers.batch_normalization default epsilon
# conv_dtype='fp8' (BASELINE config "fp8 MFMA conv path"): the convs from 128 input channels up run on the block-scaled
# e4m3 MFMA kernels; conv1_1 reads the raw fp32 input; the two 1x1 score convs read fp8 and write bf16 for the fp32
# decoder head.  conv2_1 (64 input channels) is an e4m3 conv on the generation-4 kernel, conv1_2 only with fp8_deep: see
# fp8_plan().
FP8_CONVS = ('conv2_2', 'conv3_1', 'conv3_2', 'conv3_3', 'conv4_1', 'conv4_2', 'conv4_3', 'conv5_1', 'conv5_2', 'conv5_3')
# conv outputs that may be stored as e4m3, each with a calibrated scale (pools inherit their conv's scale)
FP8_MAPS = ('conv1_1', 'conv1_2', 'conv2_1') + FP8_CONVS


# how calibrate() chooses an activation map's exponent: 'max' (largest magnitude + one bit of headroom) or 'mse' (least squared
# e4m3 error on the calibration batch; XV_FP8_CALIBRATION for A/B)
FP8_CALIBRATION = os.environ.get('XV_FP8_CALIBRATION', 'max')
FP8_DEFAULT_START = 'conv2_2'
# the accuracy-guarded plan (FcnEngine.calibrate_guarded): first e4m3 conv candidates from the deepest plan (most layers on
# e4m3 operands) to the shallowest; an expert that fails the bound on all of them runs on bf16 operands
FP8_GUARD_CANDIDATES = ('conv2_2', 'conv3_1', 'conv4_1', 'conv5_1')
FP8_GUARD_AGREEMENT = 0.995


def padded_units(num_units):
    """Lanes the `num_units`-channel decoder maps are padded to (zero weights, zero maps): 64, 128 or 256 -- the widths every
    kernel behind them takes, the batch-norm passes of training included (batchnorm.hip: C >= 64, 2048 % C == 0; a 192-lane
    map has no statistics kernel) -- and multiples of 64 beyond 256 (inference; the batch-norm trainers refuse those)."""
    u = int(num_units)
    for lanes in (64, 128, 256):
        if u <= lanes:
            return lanes
    return (u + 63) // 64 * 64


def fp8_plan(h=None, w=None, deep=False, start=None):
    """(convs with e4m3 operands, conv outputs stored as e4m3) of conv_dtype='fp8': the convs from `start` on take e4m3
    operands, the conv in front of `start` is a bf16 conv that writes the first e4m3 map.

      start='conv2_2' (default since round 5): conv1_1 fp32, conv1_2 and conv2_1 bf16, conv2_1 writes the first e4m3 map.
          tools/fp8_calib_study.py on trained experts: the depth expert (classes are thresholds on ONE raw uint16 channel, so
          its first pooled map carries the information in small relative differences) loses 2.9 - 3.7 points of mIoU when
          pool1 is stored with 3-bit mantissas and 0.01 when the first e4m3 map is conv2_1's; the RGB expert is within 0.06
          either way.  conv2_1 is 6 % of the FLOPs: ~3 % of the step at 2048x1024.
      start='conv2_1' (model config `fp8_start`; the default of rounds 2-4): conv1_2 writes the first e4m3 map (pool1).
      deep=True (model config `fp8_deep`) = start='conv1_2': conv1_1 writes the first e4m3 map: +14 % images/s at 2048x1024,
          depth expert -6.5 points.
    Any later layer up to conv5_1 may be named as well (the 1x1 score convs read conv4_3 / conv5_3 with e4m3-packed weights:
    a plan that leaves either map in bf16 would hand bf16 operands to those weights).  oracle/fcn_oracle.py states the same
    rule.  (h, w: unused since the kernels handle partial tiles; kept for callers.)"""
    names = [n for n, _, _ in ENCODER]
    start = start or ('conv1_2' if deep else FP8_DEFAULT_START)
    allowed = names[1:names.index('conv5_1') + 1]
    if start not in allowed:
        raise ValueError('fp8_start must be one of %s' % allowed)
    i = names.index(start)
    return tuple(names[i:]), tuple(names[i - 1:])


def variable_shapes(prefix, in_channels, num_units, num_classes, batch_normalization=False):
    """name -> shape of every variable of one FCN expert, in the reference npz schema
    (names: 'Synthia Rand Cityscapes Examples.ipynb':897-931; BN adds gamma/beta/moving_*)."""
    shapes = {}
    cin = in_channels
    for name, cout, _ in ENCODER:
        shapes['%s/%s/kernel' % (prefix, name)] = (3, 3, cin, cout)
        shapes['%s/%s/bias' % (prefix, name)] = (cout,)
        cin = cout
    for name in ('score_conv4', 'score_conv5'):
        shapes['%s/%s/kernel' % (prefix, name)] = (1, 1, 512, num_units)
        shapes['%s/%s/bias' % (prefix, name)] = (num_units,)
    shapes['%s/upscore_conv5/kernel' % prefix] = (4, 4, num_units, num_units)
    shapes['%s/upscore/kernel' % prefix] = (16, 16, num_units, num_units)
    shapes['%s/score/kernel' % prefix] = (1, 1, num_units, num_classes)
    shapes['%s/score/bias' % prefix] = (num_classes,)
    if batch_normalization:
        for key in list(shapes):
            if key.endswith('/kernel'):
                layer = key[:-len('/kernel')]
                c = shapes[key][2] if 'upscore' in layer else shapes[key][3]
                for v in ('gamma', 'beta', 'moving_mean', 'moving_variance'):
                    shapes['%s/%s' % (layer, v)] = (c,)
    return shapes


def init_variables(prefix, in_channels, num_units, num_classes, batch_normalization=False, seed=None):
    """[TF1] default initialisers: Glorot-uniform kernels, zero biases, bilinear deconv constants,
    BN gamma=1 beta=0 mean=0 var=1."""
    rng = np.random.default_rng(seed)
    out = {}
    for name, shape in variable_shapes(prefix, in_channels, num_units, num_classes, batch_normalization).items():
        leaf = name.rsplit('/', 1)[1]
        if 'upscore' in name and leaf == 'kernel':
            out[name] = bilinear_filter(shape)
        elif leaf == 'kernel':
            kh, kw, cin, cout = shape
            lim = np.sqrt(6.0 / (kh * kw * cin + kh * kw * cout))
            out[name] = rng.uniform(-lim, lim, size=shape).astype(np.float32)
        elif leaf in ('gamma', 'moving_variance'):
            out[name] = np.ones(shape, np.float32)
        else:
            out[name] = np.zeros(shape, np.float32)
    return out


def _fold_bn(variables, layer, kernel, bias):
    """Inference batch-norm after the conv (custom_layers.py:126-137, BN before the activation):
    y = gamma*(conv+b-mean)/sqrt(var+eps)+beta  ==  conv(x, W*s) + ((b-mean)*s+beta)."""
    g = variables.get(layer + '/gamma')
    if g is None:
        return kernel, bias
    s = g / np.sqrt(variables[layer + '/moving_variance'] + BN_EPS)
    return kernel * s, (bias - variables[layer + '/moving_mean']) * s + variables[layer + '/beta']


_FUSE_FIRST = os.environ.get('XV_FUSE_FIRST', '1') != '0'


# The two experts of a fusion model run the same layer shapes from conv1_2 on.  From this layer on each 3x3 layer of both is
# ONE launch (ops.conv2d_fwd_pair: whole rounds of workgroups where each expert alone leaves its last round half empty).
# XV_GROUP_FROM=<layer name> moves the start, XV_GROUP_FROM=0 keeps every launch per expert (A/B timing).
GROUP_FROM = os.environ.get('XV_GROUP_FROM', 'conv4_1')


def group_from_index():
    names = [e[0] for e in ENCODER]
    return names.index(GROUP_FROM) if GROUP_FROM in names else None


def encoder_layers_pair(ea, sa, eb, sb):
    """The remaining encoder layers of two engines (states from encoder_begin(..., stop=group_from_index())), each layer as
    one launch for both where the kernel takes the shape, else one launch per engine."""
    assert sa['next'] == sb['next']
    for idx in range(sa['next'], len(ENCODER)):
        name = ENCODER[idx][0]
        ya, qa = ea._layer_outputs(sa, idx)
        yb, qb = eb._layer_outputs(sb, idx)
        if not ops.conv2d_fwd_pair(sa['cur'], ea.w[name], ea.b[name], sb['cur'], eb.w[name], eb.b[name], relu=True,
                                   ya=ya, yb=yb, pa=qa, pb=qb):
            for e, st, y, q in ((ea, sa, ya, qa), (eb, sb, yb, qb)):
                ops.conv2d_fwd(st['cur'], e.w[name], e.b[name], 3, relu=True, y=y, pooled=q, write_y=y is not None)
        ea._layer_done(sa, idx, ya, qa)
        eb._layer_done(sb, idx, yb, qb)


class FcnEngine(object):
    """One FCN expert resident on one GPU (inference graph of simple_fcn.py:137-170)."""

    def __init__(self, prefix, in_channels, num_units, num_classes, variables, device='cuda', conv_dtype='bf16',
                 streamk=False, fp8_deep=False, fp8_start=None):
        self.prefix, self.cin, self.U, self.C = prefix, int(in_channels), int(num_units), int(num_classes)
        self.device = torch.device(device)
        self.Up = padded_units(self.U)            # score convs run on the MFMA kernel: pad U to 64 / 128 / 256 / ... lanes of zeros
        if conv_dtype not in ('bf16', 'fp8'):
            raise ValueError("conv_dtype must be 'bf16' or 'fp8'")
        self.conv_dtype = conv_dtype
        # streamk: hand the 3x3 convs a stream-K workspace (ops.streamk_workspace) -- the kernel then splits the tiles of an
        # incomplete last round over idle CUs where that pays (conv5_x at one image: 46 -> 27 us).  Off by default: which
        # tiles are split depends on the batch size, so an image's logits would no longer be bit-identical alone and in a
        # batch (different fp32 groupings; still bitwise reproducible run to run).  A latency option for batch 1.
        self.streamk = bool(streamk)
        self.fp8_deep = bool(fp8_deep)          # conv_dtype='fp8': e4m3 operands from conv1_2 on (see fp8_plan)
        # conv_dtype='fp8': the first conv with e4m3 operands (fp8_plan); a dict {prefix or modality: layer} picks per expert
        if isinstance(fp8_start, dict):
            fp8_start = fp8_start.get(prefix)
        # fp8_start='bf16': this expert of an fp8 model keeps bf16 operands throughout (what calibrate_guarded falls back to)
        self.fp8_off = fp8_start == 'bf16'
        self.fp8_start = None if self.fp8_off else fp8_start
        self.fp8_guard = None                     # report of the last calibrate_guarded()
        self.fp8_scales = None                    # {map name: power-of-two exponent}, set by calibrate()
        # MC dropout (simple_fcn.py:50-62,71-78,124-126; only the uncertainty models enable it): sites after which
        # tf.layers.dropout(training=True) is applied -- 'pool3', 'conv4_3', 'conv5_3' in the encoder, 'features' for the
        # decoder input -- at rate dropout_rate; every forward pass draws new masks (dropout_seed + pass counter)
        self.dropout_layers, self.dropout_rate, self.dropout_seed, self._dropout_pass = (), 0.0, 0, 0
        self._arena = {}
        self.load(variables)

    # ---- weights -----------------------------------------------------------------------------
    def load(self, variables):
        p, dev = self.prefix, self.device
        # new weights -> new activation ranges: the fp8 exponents of the previous weights would saturate (or underflow)
        # silently, so the next batch seen calibrates again (or the caller calls calibrate()).  An EXPLICIT calibration is not
        # dropped silently: the first batch after this would become the calibration set (a host sync, and exponents that
        # depend on which batch comes first -- per rank under data parallelism).
        if getattr(self, 'fp8_scales', None) is not None and getattr(self, '_fp8_explicit', False):
            import warnings
            warnings.warn('FcnEngine.load(): new weights invalidate the fp8 calibration of %r; call calibrate() again '
                          '(otherwise the next batch seen is used)' % self.prefix, RuntimeWarning, stacklevel=2)
        self.fp8_scales, self._fp8_explicit = None, False
        if getattr(self, 'fp8_guard', None) is not None:
            # a plan calibrate_guarded() chose was chosen FOR THE OLD WEIGHTS: back to the default plan until it is called again
            self.fp8_off, self.fp8_start, self.fp8_guard = False, None, None
        v = {k: np.asarray(a, np.float32) for k, a in variables.items() if k.startswith(p + '/')}
        for need, shape in variable_shapes(p, self.cin, self.U, self.C).items():
            if need not in v:
                raise KeyError('missing variable %s' % need)
            if tuple(v[need].shape) != tuple(shape):
                raise ValueError('variable %s has shape %s, expected %s' % (need, v[need].shape, shape))
        # The reference never trains its deconvs (simple_fcn.py:80-83,117-119), so their kernels are the bilinear
        # constant and run as depthwise interpolations.  An imported kernel that is anything else takes the dense
        # transposed-conv path (xv_deconv_dense_fwd on the MFMA conv; the decoder head is then the un-commuted one).
        self.dense_deconv = {}
        for name, stride in (('upscore_conv5', 2), ('upscore', 8)):
            kern = v['%s/%s/kernel' % (p, name)]
            if not is_bilinear_filter(kern):
                kp = np.zeros((kern.shape[0], kern.shape[1], self.Up, self.Up), np.float32)
                kp[:, :, :self.U, :self.U] = kern                       # padding units: zero rows and columns
                k3 = torch.from_numpy(dense_deconv_as_conv3x3(kp, stride)).to(dev)
                self.dense_deconv[name] = (ops.pack_conv_weights(k3),
                                           torch.zeros(stride * stride * self.Up, dtype=torch.float32, device=dev))
        # Batch norm after a deconv (custom_layers.py:112-119) is a per-channel affine before its relu.  A
        # positive scale with zero shift commutes with the relu and the (linear) deconv and is folded into
        # the 1x1 conv on the other side; anything else goes through the affine forms of the x2 kernel and of the
        # decoder head (the un-commuted one: 16x the FMAs of the default head).
        deconv_scale, self.affine = {}, {}
        for name in ('upscore_conv5', 'upscore'):
            layer = '%s/%s' % (p, name)
            if layer + '/gamma' in v:
                s = v[layer + '/gamma'] / np.sqrt(v[layer + '/moving_variance'] + BN_EPS)
                t = v[layer + '/beta'] - v[layer + '/moving_mean'] * s
                if np.all(s > 0) and np.all(np.abs(t) <= 1e-12) and name not in self.dense_deconv:
                    deconv_scale[name] = s.astype(np.float32)
                else:
                    sp, tp = np.ones(self.Up, np.float32), np.zeros(self.Up, np.float32)
                    sp[:self.U], tp[:self.U] = s, t                  # padding channels: relu(0 * 1 + 0) = 0
                    self.affine[name] = (torch.from_numpy(sp).to(dev), torch.from_numpy(tp).to(dev))
        self.w, self.b = {}, {}

        def up(a):
            return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)

        for name, cout, _ in ENCODER:
            k, b = _fold_bn(v, '%s/%s' % (p, name), v['%s/%s/kernel' % (p, name)], v['%s/%s/bias' % (p, name)])
            self.w[name] = up(k) if name == 'conv1_1' else ops.pack_conv_weights(up(k))
            self.b[name] = up(b)
        for name in ('score_conv4', 'score_conv5'):
            k, b = _fold_bn(v, '%s/%s' % (p, name), v['%s/%s/kernel' % (p, name)], v['%s/%s/bias' % (p, name)])
            if name == 'score_conv5' and 'upscore_conv5' in deconv_scale:
                # s*relu(up2(relu(z))) == relu(up2(relu(s*z))) for s > 0
                k, b = k * deconv_scale['upscore_conv5'], b * deconv_scale['upscore_conv5']
            kp = np.zeros((1, 1, 512, self.Up), np.float32)
            kp[..., :self.U] = k
            bp = np.zeros(self.Up, np.float32)
            bp[:self.U] = b
            self.w[name] = ops.pack_conv_weights(up(kp))
            self.b[name] = up(bp)
        k, b = _fold_bn(v, p + '/score', v[p + '/score/kernel'], v[p + '/score/bias'])
        k = k.reshape(self.U, self.C)
        if 'upscore' in deconv_scale:
            k = k * deconv_scale['upscore'][:, None]       # score(s*up) == (s-scaled score weights)(up)
        ws = np.zeros((self.Up, self.C), np.float32)
        ws[:self.U] = k
        self.w['score'] = up(ws)
        self.b['score'] = up(b)
        if self.conv_dtype == 'fp8':
            # e4m3 images of the same (batch-norm folded) kernels, each with its own power-of-two scale
            self.w8, self.w8_exp = {}, {}
            for name in ('conv1_2', 'conv2_1') + FP8_CONVS + ('score_conv4', 'score_conv5'):
                k, _ = _fold_bn(v, '%s/%s' % (p, name), v['%s/%s/kernel' % (p, name)], v['%s/%s/bias' % (p, name)])
                if name == 'score_conv5' and 'upscore_conv5' in deconv_scale:
                    k = k * deconv_scale['upscore_conv5']
                if name.startswith('score_conv'):
                    kp = np.zeros((1, 1, 512, self.Up), np.float32)
                    kp[..., :self.U] = k
                    k = kp
                self.w8[name], self.w8_exp[name] = ops.pack_conv_weights_f8(up(k))
        torch.cuda.synchronize(dev)

    def _sk(self):
        """This engine's stream-K workspace (ops.streamk_workspace: arrival counters + partial-tile slabs of the
        generation-2 conv kernel's tail round).  One per engine: the two experts of a fusion model run on two streams."""
        if not self.streamk:
            return None
        ws = self._arena.get('streamk_ws')
        if ws is None:
            ws = self._arena['streamk_ws'] = ops.streamk_workspace(self.device)
        return ws

    # ---- activations ---------------------------------------------------------------------------
    def _act(self, name, n, h, w, c, dtype='bf16', scale_exp=0):
        key = (name, n, h, w, c, dtype)
        a = self._arena.get(key)
        if a is None:
            a = ops.Act(n, h, w, c, self.device, dtype=dtype, scale_exp=scale_exp)
            self._arena[key] = a
        elif a.scale_exp != scale_exp:
            a.set_scale_exp(scale_exp)
        return a

    # ---- fp8: static per-tensor scales -----------------------------------------------------------------------
    def calibrate(self, x, margin_bits=1, _implicit=False, method=None):
        """Choose the power-of-two scale of every fp8 activation map from one representative batch: run the bf16
        graph, take max|activation| per layer, and leave `margin_bits` of headroom (an e4m3 value keeps its 3
        mantissa bits anywhere in 2^-6 .. 2^8 of the scale, so headroom costs no precision; values beyond it
        saturate at 448 * 2^e).  Returns the exponents; they stay fixed until the next call (static calibration:
        no data-dependent work on the inference path)."""
        dtype, self.conv_dtype = self.conv_dtype, 'bf16'
        try:
            L = self.encoder(x, keep_all=True)
        finally:
            self.conv_dtype = dtype
        method = method or FP8_CALIBRATION
        if method == 'mse':
            # the exponent that minimises the map's squared e4m3 error (outliers saturate), not the one that fits its maximum
            self.fp8_scales = {name: ops.fp8_scale_exp_mse(L[name].interior(), margin_bits) for name in FP8_MAPS}
        else:
            self.fp8_scales = {name: ops.fp8_scale_exp(L[name].interior().abs().max().item(), margin_bits)
                               for name in FP8_MAPS}
        self._fp8_explicit = not _implicit
        return dict(self.fp8_scales)

    def calibrate_guarded(self, x, agreement=FP8_GUARD_AGREEMENT, candidates=FP8_GUARD_CANDIDATES, margin_bits=1, method=None):
        """calibrate(), then choose this expert's e4m3 plan BY ITS EFFECT: the label map of the calibration batch through the
        bf16 graph is the yardstick, and the plan becomes the deepest candidate (first e4m3 conv as early as possible) whose
        labels agree with it on at least `agreement` of the pixels; an expert that fails the bound with every candidate keeps
        bf16 operands (fp8_off).  One global plan was a guess: tools/fp8_calib_study.py -- a weak expert (the depth expert of
        the accuracy test: thresholds on one raw uint16 channel, 0.4 % of its pixels with a clear top-2 margin) flips 1.1-1.3 %
        of its labels under ANY e4m3 plan and loses up to half a point of mIoU, the RGB expert 0.2 % and nothing.  Costs a few
        forward passes at calibration time, nothing afterwards.  Returns the exponents; self.fp8_guard holds the report
        {'chosen', 'agreement': {candidate: fraction}, 'bound'}."""
        if self.conv_dtype != 'fp8':
            raise ValueError("calibrate_guarded: conv_dtype is %r" % self.conv_dtype)
        self.fp8_off = False
        scales = self.calibrate(x, margin_bits, method=method)
        self.fp8_off = True                                        # the yardstick: this engine on bf16 operands
        ref = self.forward(x, want=('label',))['label'].clone()
        self.fp8_off = False
        report, chosen = {}, 'bf16'
        for start in candidates:
            self.fp8_start = start
            lab = self.forward(x, want=('label',))['label']
            report[start] = float((lab == ref).float().mean().item())
            if report[start] >= agreement:
                chosen = start
                break
        if chosen == 'bf16':
            self.fp8_off, self.fp8_start = True, None
        self.fp8_guard = {'chosen': chosen, 'agreement': {k: round(v, 6) for k, v in report.items()}, 'bound': float(agreement)}
        return scales

    def _encoder_fp8(self, x, keep_all=False):
        """The trunk with e4m3 operands from conv2_1 / conv2_2 on (see fp8_plan); same layer dict as `encoder`."""
        if self.fp8_scales is None:
            self.calibrate(x, _implicit=True)           # first batch seen = calibration batch
        n, h, w, _ = x.shape
        e = self.fp8_scales
        convs8, maps8 = fp8_plan(h, w, self.fp8_deep, self.fp8_start)
        L = {}
        first = 1
        ch, cw = h, w
        # conv1_1 + conv1_2 + pool1 in one launch straight onto the first e4m3 map (csrc/conv_first_fused.hip, its e4m3-out form:
        # the same bytes as the two kernels) where conv1_2 takes bf16 operands (the default plan) and neither full map is wanted
        if not keep_all and _FUSE_FIRST and 'conv1_1' not in maps8 and 'conv1_2' not in convs8 and \
                ENCODER[1][0] == 'conv1_2' and ENCODER[1][2] == 'pool1':
            # (fp8_start behind conv2_1: pool1 stays bf16)
            q = self._act('pool1', n, h // 2, w // 2, 64, **(dict(dtype='fp8', scale_exp=e['conv1_2']) if 'conv1_2' in maps8 else {}))
            if ops.conv_first_pair_fwd(x.contiguous(), self.w['conv1_1'], self.b['conv1_1'], self.w['conv1_2'],
                                       self.b['conv1_2'], pooled=q):
                L['pool1'] = cur = q
                ch, cw = h // 2, w // 2
                first = 2
        if first == 1:
            cur = self._act('conv1_1', n, h, w, 64, **(dict(dtype='fp8', scale_exp=e['conv1_1']) if 'conv1_1' in maps8 else {}))
            ops.conv2d_first_fwd(x.contiguous(), self.w['conv1_1'], self.b['conv1_1'], cur, relu=True)
            L['conv1_1'] = cur
        for name, cout, pool in ENCODER[first:]:
            f8_in = name in convs8
            f8_out = name in maps8
            wts = self.w8[name] if f8_in else self.w[name]
            kw = dict(dtype='fp8', scale_exp=e[name]) if f8_out else {}
            if pool is None:
                y = self._act(name, n, ch, cw, cout, **kw)
                ops.conv2d_fwd(cur, wts, self.b[name], 3, relu=True, y=y)
                L[name] = cur = y
            else:
                q = self._act(pool, n, ch // 2, cw // 2, cout, **kw)
                need_full = keep_all or name == 'conv4_3'
                y = self._act(name, n, ch, cw, cout, **kw) if need_full else None
                ops.conv2d_fwd(cur, wts, self.b[name], 3, relu=True, y=y, pooled=q, write_y=need_full)
                if y is not None:
                    L[name] = y
                L[pool] = cur = q
                ch, cw = ch // 2, cw // 2
        s4 = self._act('score_conv4', n, h // 8, w // 8, self.Up)
        ops.conv2d_fwd(L['conv4_3'], self.w8['score_conv4'], self.b['score_conv4'], 1, relu=True, y=s4)
        s5 = self._act('score_conv5', n, h // 16, w // 16, self.Up)
        ops.conv2d_fwd(L['conv5_3'], self.w8['score_conv5'], self.b['score_conv5'], 1, relu=True, y=s5)
        return L, s4, s5

    def encoder(self, x, keep_all=False, routed=False):
        """x: float32 [N,H,W,cin] device tensor (raw 0..255 RGB / raw depth, data contract of
        xview/datasets/*) -> dict of Acts; 'fused' is the encoding (simple_fcn.py:10-87).
        keep_all=True also materialises every convX_Y / poolX like the reference's layer dict.
        routed=True (with keep_all; the training step): a conv in front of a pool whose full map only MaxPoolGrad would read
        leaves 'route_<name>' (uint8 route bytes, ops.conv2d_fwd_route) in the dict instead of '<name>', where the kernel
        takes the shape."""
        return self.encoder_finish(self.encoder_begin(x, keep_all=keep_all, stop=len(ENCODER), routed=routed))

    # The encoder in three steps, so that a fusion model can run the layers its two experts share as ONE launch each
    # (encoder_layers_pair below): encoder_begin (input checks, the first conv(s), the layers before `stop`), the remaining
    # layers (here or paired), encoder_finish (score convs, x2 upsampling + skip).
    def encoder_begin(self, x, keep_all=False, stop=None, routed=False):
        n, h, w, cin = x.shape
        if cin != self.cin:
            raise ValueError('expected %d input channels, got %d' % (self.cin, cin))
        if h % 16 or w % 16:
            raise ValueError('H and W must be multiples of 16 (augmentation.py:244-262 crop_multiple)')
        st = {'n': n, 'h': h, 'w': w, 'keep_all': keep_all, 'routed': bool(routed) and keep_all}
        if self.conv_dtype == 'fp8' and not self.fp8_off:
            st['L'], st['s4'], st['s5'] = self._encoder_fp8(x, keep_all)
            st['next'] = len(ENCODER)
            return st
        L = {}
        ch, cw = h, w
        first = 1
        # conv1_1 + conv1_2 + pool1 in one launch where neither full-resolution map is wanted (inference): conv1_1 is
        # evaluated straight into conv1_2's LDS patch buffers (csrc/conv_first_fused.hip; the same bits as the two
        # kernels).  Maps that do not tile in 16x32 -- and XV_FUSE_FIRST=0, A/B timing -- take the two kernels.
        if not keep_all and _FUSE_FIRST and ENCODER[1][0] == 'conv1_2' and ENCODER[1][2] == 'pool1':
            q = self._act('pool1', n, h // 2, w // 2, 64)
            if ops.conv_first_pair_fwd(x.contiguous(), self.w['conv1_1'], self.b['conv1_1'], self.w['conv1_2'],
                                       self.b['conv1_2'], pooled=q):
                L['pool1'] = cur = q
                ch, cw = h // 2, w // 2
                first = 2
        if first == 1:
            cur = self._act('conv1_1', n, h, w, 64)
            ops.conv2d_first_fwd(x.contiguous(), self.w['conv1_1'], self.b['conv1_1'], cur, relu=True)
            L['conv1_1'] = cur
        st.update(L=L, cur=cur, ch=ch, cw=cw, next=first)
        self.encoder_layers(st, len(ENCODER) if stop is None else stop)
        return st

    def _drop_fn(self):
        return self._dropout if self.dropout_layers and self.dropout_rate > 0 else None

    def _layer_outputs(self, st, idx):
        """(full map or None, pooled map or None) of encoder layer idx"""
        name, cout, pool = ENCODER[idx]
        n, ch, cw = st['n'], st['ch'], st['cw']
        if pool is None:
            return self._act(name, n, ch, cw, cout), None
        need_full = st['keep_all'] or name == 'conv4_3'       # conv4_3 feeds score_conv4
        return (self._act(name, n, ch, cw, cout) if need_full else None), self._act(pool, n, ch // 2, cw // 2, cout)

    def _layer_done(self, st, idx, y, q):
        name, cout, pool = ENCODER[idx]
        L = st['L']
        if pool is None:
            L[name] = st['cur'] = y
        else:
            if y is not None:
                L[name] = y
            L[pool] = st['cur'] = q
            st['ch'], st['cw'] = st['ch'] // 2, st['cw'] // 2
            # simple_fcn.py:51-63: the dropout after pool4 is gated by 'pool3' too (a quirk of the reference:
            # 'pool4' alone enables nothing)
            drop = self._drop_fn()
            if drop is not None and pool in ('pool3', 'pool4') and 'pool3' in self.dropout_layers:
                L[pool + '_drop'] = st['cur'] = drop(q, pool + '_drop')
        st['next'] = idx + 1

    def encoder_layers(self, st, stop):
        """encoder layers [st['next'], stop), one launch each"""
        for idx in range(st['next'], stop):
            name = ENCODER[idx][0]
            y, q = self._layer_outputs(st, idx)
            if st.get('routed') and q is not None and name != 'conv4_3' and self._drop_fn() is None:
                key = ('route_' + name, st['n'], q.h, q.w, q.c)
                route = self._arena.get(key)
                if route is None:
                    route = self._arena[key] = torch.empty(st['n'] * q.h * q.w * q.c, dtype=torch.uint8, device=self.device)
                if ops.conv2d_fwd_route(st['cur'], self.w[name], self.b[name], q, route):
                    st['L']['route_' + name] = route
                    self._layer_done(st, idx, None, q)
                    continue
            ops.conv2d_fwd(st['cur'], self.w[name], self.b[name], 3, relu=True, y=y, pooled=q, write_y=y is not None,
                           workspace=self._sk(),
                           split_ws=ops.split_workspace(st['cur'], ENCODER[idx][1], self._arena) if q is None else None)
            self._layer_done(st, idx, y, q)

    def pairable(self):
        """May this engine's 3x3 layers share launches with a twin's (encoder_layers_pair)?"""
        return self.conv_dtype == 'bf16' and self._drop_fn() is None and self._sk() is None

    def encoder_finish(self, st):
        n, h, w = st['n'], st['h'], st['w']
        L = st['L']
        if 's4' in st:
            s4, s5 = st['s4'], st['s5']
        else:
            self.encoder_layers(st, len(ENCODER))
            drop = self._drop_fn()
            s4 = self._act('score_conv4', n, h // 8, w // 8, self.Up)
            c43 = drop(L['conv4_3'], 'conv4_3_drop') if drop is not None and 'conv4_3' in self.dropout_layers else L['conv4_3']
            ops.conv2d_fwd(c43, self.w['score_conv4'], self.b['score_conv4'], 1, relu=True, y=s4)
            s5 = self._act('score_conv5', n, h // 16, w // 16, self.Up)
            c53 = drop(L['conv5_3'], 'conv5_3_drop') if drop is not None and 'conv5_3' in self.dropout_layers else L['conv5_3']
            ops.conv2d_fwd(c53, self.w['score_conv5'], self.b['score_conv5'], 1, relu=True, y=s5)
        fused = self._act('fused', n, h // 8, w // 8, self.Up)
        aff = self.affine.get('upscore_conv5', (None, None))
        if 'upscore_conv5' in self.dense_deconv:
            wk, zb = self.dense_deconv['upscore_conv5']
            _, self._arena['dd_ws5'] = ops.deconv_dense_fwd(s5, wk, zb, 2, self.Up, y=fused, scale=aff[0], shift=aff[1],
                                                           residual=s4, relu=True, workspace=self._arena.get('dd_ws5'))
        else:
            ops.upsample2x_relu_add(s5, residual=s4, y=fused, scale=aff[0], shift=aff[1])
        L.update(score_conv4=s4, score_conv5=s5, fused=fused)
        if self.dropout_rate > 0 and 'features' in self.dropout_layers:
            # decoder(..., dropout_rate) drops its input features INSIDE the decoder (simple_fcn.py:124-126): the layer
            # dict keeps 'fused' undropped like the reference's, the head reads 'features_drop'
            L['features_drop'] = self._dropout(fused, 'features_drop')
        self._dropout_pass += 1
        return L

    def set_dropout(self, dropout_layers, dropout_rate, seed=0):
        """Enable (or, with an empty list / rate 0, disable) the MC-dropout sites of encoder / decoder."""
        if self.conv_dtype != 'bf16' and dropout_layers and dropout_rate > 0:
            raise NotImplementedError("dropout sites exist in the bf16 graph only (conv_dtype='fp8' is plain inference)")
        self.dropout_layers, self.dropout_rate, self.dropout_seed = tuple(dropout_layers), float(dropout_rate), int(seed)

    def _dropout(self, x, tag):
        site = sum(ord(ch) for ch in tag)
        seed = (self.dropout_seed * 0x9E3779B1 + self._dropout_pass * 1000003 + site) & 0xffffffffffffffff
        return ops.dropout(x, self.dropout_rate, seed, y=self._act(tag, x.n, x.h, x.w, x.c))

    def commuted_head(self):
        """True if the decoder head runs in its commuted form (class scores interpolated at 1/8 resolution): no batch-norm
        shift between the x8 deconv and its relu, bilinear deconv kernel -- the precondition of `lowres_scores`."""
        return 'upscore' not in self.affine and 'upscore' not in self.dense_deconv

    def lowres_scores(self, x=None, st=None):
        """Trunk + the 1x1 score conv at 1/8 resolution: float32 [N][h/8+2][w/8+2][CP] (the first half of the decoder
        head, xv_score_lowres); the fused two-expert head of the fusion models takes it from here.  st: an encoder state
        (encoder_begin / encoder_layers_pair) to finish instead of an input."""
        L = self.encoder_finish(st) if st is not None else self.encoder(x)
        f = L.get('features_drop', L['fused'])
        cp = (self.C + 3) // 4 * 4
        key = ('lowres_S', f.n, f.h, f.w)
        S = self._arena.get(key)
        if S is None:
            S = self._arena[key] = torch.zeros((f.n, f.h + 2, f.w + 2, cp), dtype=torch.float32, device=self.device)
        ops.score_lowres(f, self.w['score'], self.C, S)
        return S, (f.n, f.h, f.w)

    def forward(self, x, want=('label',), keep_all=False, st=None):
        """fcn + test_pipeline (basic_fusion_model.py:9-23): returns dict with any of
        'score', 'prob' (float32 [N,H,W,C]) and 'label' == 'classification' (int64 [N,H,W])."""
        if st is not None and bool(keep_all) != bool(st['keep_all']):
            raise ValueError('forward(st=...): the encoder state was begun with keep_all=%r' % st['keep_all'])
        L = self.encoder_finish(st) if st is not None else self.encoder(x, keep_all=keep_all)
        f = L.get('features_drop', L['fused'])           # the decoder's input (dropped only when 'features' is a dropout site)
        key = ('head_ws', f.n, f.h, f.w)
        ws = self._arena.get(key)
        if ws is None:             # per-engine workspace: the two experts run on different streams
            ws = torch.empty(ops._lib.lib().xv_decoder_head_workspace_bytes(f.n, f.h, f.w, self.C) // 4,
                             dtype=torch.float32, device=self.device)
            self._arena[key] = ws
        aff = self.affine.get('upscore', (None, None))
        if 'upscore' in self.dense_deconv:
            # dense x8 deconv -> [batch norm] -> relu at full resolution, then the per-pixel score conv, softmax, argmax
            wk, zb = self.dense_deconv['upscore']
            up = self._act('upscore', f.n, 8 * f.h, 8 * f.w, self.Up)
            _, self._arena['dd_ws'] = ops.deconv_dense_fwd(f, wk, zb, 8, self.Up, y=up, scale=aff[0], shift=aff[1],
                                                          relu=True, workspace=self._arena.get('dd_ws'))
            skey = ('dense_score', f.n, f.h, f.w)
            if skey not in self._arena:
                self._arena[skey] = torch.empty((f.n, 8 * f.h, 8 * f.w, self.C), dtype=torch.float32, device=self.device)
            score = ops.score_dense_fwd(up, self.w['score'], self.b['score'], self.C, self._arena[skey])
            want_label = 'label' in want or 'classification' in want
            prob, label = ops.softmax_argmax(score, want_prob='prob' in want, want_label=want_label)
            out = {'layers': L}
            L['upscore'] = up
            if 'score' in want:
                out['score'] = score
            if prob is not None:
                out['prob'] = prob
            if label is not None:
                out['label'] = out['classification'] = label
            return out
        out = ops.decoder_head_fwd(f, self.w['score'], self.b['score'], self.C,
                                   want_score='score' in want, want_prob='prob' in want,
                                   want_label=('label' in want or 'classification' in want), workspace=ws,
                                   scale=aff[0], shift=aff[1])
        if 'label' in out:
            out['classification'] = out['label']
        out['layers'] = L
        return out
