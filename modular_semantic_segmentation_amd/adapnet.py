"""The AdapNet expert on MI355X (inference graph): host-side mirror of `block_a` / `block_b` / `adapnet` / `Adapnet`
(xview/models/adapnet.py:12-219) over the same HIP kernels as the FCN expert.

Every conv of the graph is followed by a batch norm and (mostly) a relu (custom_layers.py:124-139); at inference
the batch norm folds into kernel and bias, so the trunk is 1x1 and 3x3 stride-1 convs on the MFMA kernels plus three
gathers (csrc/resnet_ops.hip) that express the rest through them:
  * block_0_2, 7x7 stride 2      -> xv_gather_conv7s2 + ONE 3x3 conv over 9*64 channels, 2x2 max-pool fused (inference:
    block_0_1's kernel writes that operand itself, xv_conv2d_first_gather7s2_fwd);
  * 1x1 stride 2 (blocks 4, 8)   -> xv_subsample2 + 1x1 conv;
  * block_b's two atrous 3x3     -> ONE implicit GEMM over nine taps per output half whose output is already their concat
    (xv_conv_dilated_pair_fwd; halves that are not multiples of 128 channels, block_layer_7: xv_im2col_dilated_pair + a
    1x1 conv over the 18C operand, the same bits);
  * stage_3 + shortcut + relu    -> xv_conv2d_fwd_residual (the outer relu acts on a sum of two relu outputs);
  * the two deconvs (kernel [k,k,filters,in], custom_layers.py:71-121).  The reference leaves them TRAINABLE
    (adapnet.py:155-163 omits trainable=False), so a kernel is either still the bilinear constant it is initialised to
    -- non-zero only at [.,.,i,i], evaluated as channel slices: `first_deconvolution_conv` computes only the num_units
    channels the x2 deconv reads, and the x8 deconv + batch norm + softmax + argmax is the FCN decoder-head kernel with
    a diagonal "score conv" holding the batch-norm scale -- or dense (any trained checkpoint), evaluated as ONE 3x3 MFMA
    conv onto stride^2 phase channels at the input resolution + a depth-to-space shuffle (custom_layers.
    dense_deconv_as_conv3x3; xv_deconv_dense_fwd / xv_depth_to_space_dense).
Training: adapnet_trainer.AdapnetTrainer (every batch norm in training mode, the gathers' transposes, both deconv
kernels trained as the reference trains them).
"""
import os

import numpy as np
import torch

from . import ops
from .base_model import BaseModel
from .custom_layers import bilinear_filter, dense_deconv_as_conv3x3
from .fcn import BN_EPS

# (scope, kind, arguments): a = (intermediate, filters, stride, shortcut_conv),
# b = (filters_1, filters_2, filters_3, dilation1, dilation2, shortcut_conv)           adapnet.py:130-155
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
BN_VARS = ('gamma', 'beta', 'moving_mean', 'moving_variance')


def _conv_scopes(in_channels, num_units, blocks=None):
    """(scope, k, cin, cout, has_bias) of every conv; blocks pass use_bias=False, the rest keep the default bias."""
    scopes = [('block_0_1', 3, in_channels, 64, True), ('block_0_2', 7, 64, 64, True)]
    cin = 64
    for name, kind, args in (blocks or BLOCKS):
        if kind == 'a':
            mid, cout, _, shortcut = args
            scopes += [(name + '/stage_1', 1, cin, mid, False), (name + '/stage_2', 3, mid, mid, False),
                       (name + '/stage_3', 1, mid, cout, False)]
        else:
            f1, f2, cout, _, _, shortcut = args
            scopes += [(name + '/stage_1', 1, cin, f1, False), (name + '/stage_2_1', 3, f1, f2 // 2, False),
                       (name + '/stage_2_2', 3, f1, f2 // 2, False), (name + '/stage_3', 1, f2, cout, False)]
        if shortcut:
            scopes.append((name + '/shortcut', 1, cin, cout, False))
        if name == 'block_layer_7':
            scopes.append(('shortcut', 1, cout, num_units, True))
        cin = cout
    scopes.append(('first_deconvolution_conv', 1, cin, cin, True))      # 2048 -> 2048 in the reference graph
    return scopes


def variable_shapes(prefix, in_channels, num_units, num_classes, blocks=None):
    """name -> shape in the reference's npz schema: a conv and its batch norm share one variable scope."""
    shapes = {}
    scopes = _conv_scopes(in_channels, num_units, blocks)
    for scope, k, cin, cout, has_bias in scopes:
        shapes['%s/%s/kernel' % (prefix, scope)] = (k, k, cin, cout)
        if has_bias:
            shapes['%s/%s/bias' % (prefix, scope)] = (cout,)
        for v in BN_VARS:
            shapes['%s/%s/%s' % (prefix, scope, v)] = (cout,)
    for scope, k, filters, cin in (('first_deconvolution_upconv', 4, num_units, scopes[-1][3]),
                                   ('second_deconvolution_upconv', 16, num_classes, num_units)):
        shapes['%s/%s/kernel' % (prefix, scope)] = (k, k, filters, cin)
        for v in BN_VARS:
            shapes['%s/%s/%s' % (prefix, scope, v)] = (filters,)
    return shapes


def rect_bilinear_filter(shape):
    """bilinear_filter_initializer for the [k,k,filters,in] kernels AdapNet asks for (custom_layers.py:8-25)."""
    return bilinear_filter(tuple(shape))


def init_variables(prefix, in_channels, num_units, num_classes, seed=None, blocks=None):
    """[TF1] default initialisers: Glorot-uniform kernels, zero biases, gamma 1, beta 0, mean 0, variance 1."""
    rng = np.random.default_rng(seed)
    out = {}
    for name, shape in variable_shapes(prefix, in_channels, num_units, num_classes, blocks).items():
        leaf = name.rsplit('/', 1)[1]
        if 'deconvolution_upconv' in name and leaf == 'kernel':
            out[name] = rect_bilinear_filter(shape)
        elif leaf == 'kernel':
            kh, kw, cin, cout = shape
            lim = np.sqrt(6.0 / (kh * kw * cin + kh * kw * cout))
            out[name] = rng.uniform(-lim, lim, size=shape).astype(np.float32)
        elif leaf in ('gamma', 'moving_variance'):
            out[name] = np.ones(shape, np.float32)
        else:
            out[name] = np.zeros(shape, np.float32)
    return out


def conv7s2_as_3x3(kernel):
    """[7,7,C,Co] stride-2 'same' kernel -> the [3,3,9C,Co] stride-1 kernel over the xv_gather_conv7s2 operand.
    Per axis the variants (parity p, shift s) = (0,0), (0,1), (1,0) carry taps k = 2(t+s)+p; the shifted variant
    only supplies k = 6 (t = 2)."""
    _, _, c, co = kernel.shape
    variants = ((0, 0), (0, 1), (1, 0))
    out = np.zeros((3, 3, 9 * c, co), np.float32)
    for rv, (pr, sr) in enumerate(variants):
        for cv, (pc, sc) in enumerate(variants):
            g = rv * 3 + cv
            for tr in range(3):
                if sr and tr != 2:
                    continue
                for tc in range(3):
                    if sc and tc != 2:
                        continue
                    out[tr, tc, g * c:(g + 1) * c] = kernel[2 * (tr + sr) + pr, 2 * (tc + sc) + pc]
    return out


def dilated_pair_as_1x1(kernel1, kernel2):
    """Two [3,3,C,F/2] atrous kernels -> the [1,1,18C,F] kernel over the xv_im2col_dilated_pair operand; output channels
    [0,F/2) see only the first nine taps, [F/2,F) only the last nine: tf.concat([stage_2_1, stage_2_2]) in one conv."""
    _, _, c, half = kernel1.shape
    out = np.zeros((1, 1, 18 * c, 2 * half), np.float32)
    out[0, 0, :9 * c, :half] = kernel1.reshape(9 * c, half)
    out[0, 0, 9 * c:, half:] = kernel2.reshape(9 * c, half)
    return out


class AdapnetEngine(object):
    """One AdapNet expert resident on one GPU; same surface as fcn.FcnEngine (load / forward)."""

    def __init__(self, prefix, in_channels, num_units, num_classes, variables, device='cuda', blocks=None):
        """blocks: another block list in the format of BLOCKS (tests run shallow graphs); default: the reference's 16."""
        self.prefix, self.cin, self.U, self.C = prefix, int(in_channels), int(num_units), int(num_classes)
        self.blocks = list(blocks or BLOCKS)
        if self.C > self.U:
            raise ValueError('the x8 deconv kernel [16,16,num_classes,num_units] needs num_classes <= num_units '
                             '(custom_layers.py:20-24)')
        self.device = torch.device(device)
        self.Up = ((self.U + 63) // 64) * 64
        self._arena = {}
        self.implicit_pairs = os.environ.get('XV_IMPLICIT_PAIRS', '1') != '0'   # 0: the materialised form everywhere (A/B)
        self.fused_first = os.environ.get('XV_ADAPNET_FUSED_FIRST', '1') != '0'  # 0: block_0_1's map written, then gathered
        self.load(variables)

    # ---- weights -------------------------------------------------------------------------------------------
    def load(self, variables):
        p, dev = self.prefix, self.device
        v = {k: np.asarray(a, np.float32) for k, a in variables.items() if k.startswith(p + '/')}
        for need, shape in variable_shapes(p, self.cin, self.U, self.C, self.blocks).items():
            if need not in v:
                raise KeyError('missing variable %s' % need)
            if tuple(v[need].shape) != tuple(shape):
                raise ValueError('variable %s has shape %s, expected %s' % (need, v[need].shape, shape))
        # which of the two deconv kernels is still the bilinear constant (fast depthwise forms) and which is dense
        self.dense = {}
        dense_scopes = [scope for scope in ('first_deconvolution_upconv', 'second_deconvolution_upconv')
                        if not np.allclose(v['%s/%s/kernel' % (p, scope)],
                                           rect_bilinear_filter(v['%s/%s/kernel' % (p, scope)].shape), atol=1e-6)]

        def up(a):
            return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)

        def affine(scope):
            s = v['%s/%s/gamma' % (p, scope)] / np.sqrt(v['%s/%s/moving_variance' % (p, scope)] + BN_EPS)
            return s, v['%s/%s/beta' % (p, scope)] - v['%s/%s/moving_mean' % (p, scope)] * s

        def folded(scope):
            """conv -> batch norm == conv with kernel * s and bias b * s + t."""
            s, t = affine(scope)
            b = v.get('%s/%s/bias' % (p, scope))
            return v['%s/%s/kernel' % (p, scope)] * s, (t if b is None else b * s + t)

        self.w, self.b = {}, {}
        k, b = folded('block_0_1')
        self.w['block_0_1'], self.b['block_0_1'] = up(k), up(b)            # fp32 first-layer kernel
        k, b = folded('block_0_2')
        self.w['block_0_2'], self.b['block_0_2'] = ops.pack_conv_weights(up(conv7s2_as_3x3(k))), up(b)
        for name, kind, args in self.blocks:
            stages = ['stage_1', 'stage_3'] + (['stage_2'] if kind == 'a' else []) + (['shortcut'] if args[-1] else [])
            for stage in stages:
                k, b = folded('%s/%s' % (name, stage))
                self.w[name + '/' + stage], self.b[name + '/' + stage] = ops.pack_conv_weights(up(k)), up(b)
            if kind == 'b':
                k1, b1 = folded(name + '/stage_2_1')
                k2, b2 = folded(name + '/stage_2_2')
                self.w[name + '/stage_2'] = ops.pack_conv_weights(up(dilated_pair_as_1x1(k1, k2)))
                self.b[name + '/stage_2'] = up(np.concatenate([b1, b2]))

        def padded_units(kernel, bias):
            kp = np.zeros(kernel.shape[:3] + (self.Up,), np.float32)
            kp[..., :self.U] = kernel[..., :self.U]
            bp = np.zeros(self.Up, np.float32)
            bp[:self.U] = bias[:self.U]
            return ops.pack_conv_weights(up(kp)), up(bp)

        self.w['shortcut'], self.b['shortcut'] = padded_units(*folded('shortcut'))
        s, t = affine('first_deconvolution_upconv')
        sp, tp = np.ones(self.Up, np.float32), np.zeros(self.Up, np.float32)
        sp[:self.U], tp[:self.U] = s, t
        self.deconv1_affine = (up(sp), up(tp))
        if 'first_deconvolution_upconv' in dense_scopes:
            # trained x2 deconv [4,4,U,Cin]: every channel of first_deconvolution_conv feeds it
            k, b = folded('first_deconvolution_conv')
            self.w['first_deconvolution_conv'], self.b['first_deconvolution_conv'] = ops.pack_conv_weights(up(k)), up(b)
            w1 = v['%s/first_deconvolution_upconv/kernel' % p]
            kp = np.zeros((4, 4, self.Up, w1.shape[3]), np.float32)
            kp[:, :, :self.U] = w1
            self.dense['first'] = (ops.pack_conv_weights(up(dense_deconv_as_conv3x3(kp, 2))),
                                   torch.zeros(4 * self.Up, dtype=torch.float32, device=dev), int(w1.shape[3]))
        else:
            # bilinear constant: the x2 deconv reads only channels [0, num_units) of first_deconvolution_conv
            self.w['first_deconvolution_conv'], self.b['first_deconvolution_conv'] = \
                padded_units(*folded('first_deconvolution_conv'))
        s, t = affine('second_deconvolution_upconv')
        if 'second_deconvolution_upconv' in dense_scopes:
            # trained x8 deconv [16,16,C,U]: 3x3 conv U -> 64 phases x Cp classes at 1/8 resolution, then the shuffle
            # (with the batch norm's scale / shift) straight into the dense float32 class scores
            w2 = v['%s/second_deconvolution_upconv/kernel' % p]
            self.Cp = (self.C + 7) // 8 * 8
            kp = np.zeros((16, 16, self.Cp, self.Up), np.float32)
            kp[:, :, :self.C, :self.U] = w2
            # (the derived 3x3 kernel stays float32: the class scores are computed in float32, ops.deconv8_scores_f32)
            self.dense['second'] = (up(dense_deconv_as_conv3x3(kp, 8)), None, up(s), up(t))
        # bilinear constant: score = bilinear_x8(merge[..., :C]) * s + t, a diagonal score "conv" in front of the
        # interpolation (kept in both cases: the trainer reads its shapes)
        ws = np.zeros((self.Up, self.C), np.float32)
        ws[np.arange(self.C), np.arange(self.C)] = s
        self.w['score'], self.b['score'] = up(ws), up(t)
        torch.cuda.synchronize(dev)

    # ---- activations -----------------------------------------------------------------------------------------
    def _act(self, name, n, h, w, c):
        key = (name, n, h, w, c)
        a = self._arena.get(key)
        if a is None:
            a = ops.Act(n, h, w, c, self.device)
            self._arena[key] = a
        return a

    def _conv(self, name, x, k, cout, relu=True):
        y = self._act(name, x.n, x.h, x.w, cout)
        ops.conv2d_fwd(x, self.w[name], self.b[name], k, relu=relu, y=y)
        return y

    def trunk(self, x, keep_all=True):
        """x: float32 [N,H,W,cin] device tensor -> dict of Acts (keys as the reference's layer dict, adapnet.py:165-173;
        block_0_2 itself is only produced pooled; keep_all=False: 'block_0_1' is not produced either where its kernel can
        write the operand of block_0_2 directly)."""
        n, h, w, cin = x.shape
        if cin != self.cin:
            raise ValueError('expected %d input channels, got %d' % (self.cin, cin))
        if h % 16 or w % 16:
            raise ValueError('H and W must be multiples of 16 (augmentation.py:244-262 crop_multiple)')
        L = {}
        z = self._act('block_0_2/operand', n, h // 2, w // 2, 9 * 64)
        if not keep_all and self.fused_first and cin in (1, 3):
            # block_0_1 straight into the operand of block_0_2: its 64-channel full-resolution map is neither written nor read
            ops.conv2d_first_gather7s2_fwd(x.contiguous(), self.w['block_0_1'], self.b['block_0_1'], z, relu=True)
        else:
            cur = self._act('block_0_1', n, h, w, 64)
            ops.conv2d_first_fwd(x.contiguous(), self.w['block_0_1'], self.b['block_0_1'], cur, relu=True)
            L['block_0_1'] = cur
            ops.gather_conv7s2(cur, z)
        cur = self._act('block_0_pool', n, h // 4, w // 4, 64)
        ops.conv2d_fwd(z, self.w['block_0_2'], self.b['block_0_2'], 3, relu=True, pooled=cur, write_y=False)
        L['block_0_pool'] = cur
        for index, (name, kind, args) in enumerate(self.blocks, start=1):
            inp = cur
            if kind == 'a':
                mid, cout, stride, shortcut_conv = args
                if stride == 2:
                    inp = ops.subsample2(cur, self._act(name + '/input', cur.n, cur.h // 2, cur.w // 2, cur.c))
                s1 = self._conv(name + '/stage_1', inp, 1, mid)
                s2 = self._conv(name + '/stage_2', s1, 3, mid)
            else:
                f1, f2, cout, d1, d2, shortcut_conv = args
                s1 = self._conv(name + '/stage_1', inp, 1, f1)
                if self.implicit_pairs and ops.dilated_pair_implicit_ok(f1, f2):
                    # the nine taps of each half gathered by the GEMM's own loads: no 18 f1 operand (ops.conv_dilated_pair)
                    s2 = ops.conv_dilated_pair(s1, self.w[name + '/stage_2'], self.b[name + '/stage_2'], d1, d2, relu=True,
                                               y=self._act(name + '/stage_2', s1.n, s1.h, s1.w, f2))
                else:
                    z = ops.im2col_dilated_pair(s1, d1, d2, self._act(name + '/operand', s1.n, s1.h, s1.w, 18 * f1))
                    s2 = self._conv(name + '/stage_2', z, 1, f2)
            short = self._conv(name + '/shortcut', inp, 1, cout) if shortcut_conv else inp
            cur = ops.conv1x1_residual(s2, self.w[name + '/stage_3'], self.b[name + '/stage_3'], short, relu=True,
                                       y=self._act('block_%d' % index, s2.n, s2.h, s2.w, cout))
            L['block_%d' % index] = cur
            if name == 'block_layer_7':
                L['shortcut'] = self._conv('shortcut', cur, 1, self.Up, relu=False)
        if 'first' in self.dense:
            wk, zb, width = self.dense['first']
            d = self._conv('first_deconvolution_conv', cur, 1, width)
            L['merge'], self._arena['dd_ws1'] = ops.deconv_dense_fwd(
                d, wk, zb, 2, self.Up, y=self._act('merge', d.n, 2 * d.h, 2 * d.w, self.Up), scale=self.deconv1_affine[0],
                shift=self.deconv1_affine[1], residual=L['shortcut'], relu=False, workspace=self._arena.get('dd_ws1'))
            return L
        d = self._conv('first_deconvolution_conv', cur, 1, self.Up)
        L['merge'] = ops.upsample2x_relu_add(d, residual=L['shortcut'], scale=self.deconv1_affine[0],
                                             shift=self.deconv1_affine[1], relu=False,
                                             y=self._act('merge', d.n, 2 * d.h, 2 * d.w, self.Up))
        return L

    def forward(self, x, want=('label',), keep_all=False):
        """adapnet + test_pipeline (basic_fusion_model.py:9-23): any of 'score', 'prob' (float32 [N,H,W,C]) and
        'label' == 'classification' (int64 [N,H,W])."""
        L = self.trunk(x, keep_all=keep_all)
        m = L['merge']
        if 'second' in self.dense:
            wk, _, sc, sh = self.dense['second']
            skey = ('dense_score', m.n, m.h, m.w)
            if skey not in self._arena:
                self._arena[skey] = torch.empty((m.n, 8 * m.h, 8 * m.w, self.C), dtype=torch.float32, device=self.device)
            score = ops.deconv8_scores_f32(m, wk, self.C, self.Cp, self._arena[skey], self._arena, scale=sc, shift=sh)
            want_label = 'label' in want or 'classification' in want
            prob, label = (ops.softmax_argmax(score, want_prob='prob' in want, want_label=want_label)
                           if ('prob' in want or want_label) else (None, None))
            out = {'layers': L}
            if 'score' in want:
                out['score'] = score
            if prob is not None:
                out['prob'] = prob
            if label is not None:
                out['label'] = out['classification'] = label
            return out
        key = ('head_ws', m.n, m.h, m.w)
        ws = self._arena.get(key)
        if ws is None:
            ws = torch.empty(ops._lib.lib().xv_decoder_head_workspace_bytes(m.n, m.h, m.w, self.C) // 4,
                             dtype=torch.float32, device=self.device)
            self._arena[key] = ws
        out = ops.decoder_head_fwd(m, self.w['score'], self.b['score'], self.C, want_score='score' in want,
                                   want_prob='prob' in want,
                                   want_label=('label' in want or 'classification' in want), workspace=ws)
        if 'label' in out:
            out['classification'] = out['label']
        out['layers'] = L
        return out


class Adapnet(BaseModel):
    """AdapNet expert behind the BaseModel API (adapnet.py:175-219): `Adapnet(data_description, prefix=None,
    output_dir=None, **config)` with config keys modality, num_units (and num_classes from the data description),
    optional trainer / learning_rate / batchsize."""

    def __init__(self, data_description, prefix=None, output_dir=None, **config):
        standard_config = {'train_encoder': True}
        standard_config.update(config)
        self.prefix = config['modality'] if prefix is None else prefix
        self.modality = config['modality']
        BaseModel.__init__(self, data_description, output_dir=output_dir, **standard_config)

    def _build_graph(self):
        self.in_channels = int(self.testdata_description[1][self.modality][-1])
        blocks = self.config.get('blocks')          # None = the reference's 16 blocks; tests pass shallower lists
        self.variables = init_variables(self.prefix, self.in_channels, self.config['num_units'],
                                        self.config['num_classes'], seed=self.config.get('seed'), blocks=blocks)
        self.engine = AdapnetEngine(self.prefix, self.in_channels, self.config['num_units'],
                                    self.config['num_classes'], self.variables, device=self.device, blocks=blocks)
        self.loss = None
        self.prediction = 'label'

    def _variables_changed(self):
        BaseModel._variables_changed(self)
        self.engine.load(self.variables)
        if getattr(self, 'trainer', None) is not None:
            from .parallel import sync_trainer_from_rank0
            self.trainer.load_from_variables(self.variables)
            self._after_rank0_sync(sync_trainer_from_rank0(self.trainer))

    # ---- training (adapnet.py:190-203, optimizer setup base_model.py:153-162) ---------------------------------------
    def _ensure_trainer(self):
        if getattr(self, 'trainer', None) is None:
            from .adapnet_trainer import AdapnetTrainer
            from .parallel import GradReducer, require_equal_batchsize, sync_trainer_from_rank0, world
            self.trainer = AdapnetTrainer(self.engine, self.config.get('trainer', 'adam'),
                                          self.config.get('learning_rate', 0.0001))
            self.trainer.load_from_variables(self.variables)
            # every replica starts from rank 0's initialisers
            self._after_rank0_sync(sync_trainer_from_rank0(self.trainer))
            require_equal_batchsize(self.config['batchsize'], self.device)
            self._reducer = GradReducer(self.device) if world()[1] > 1 else None
        return self.trainer

    def _after_rank0_sync(self, synced):
        """Rank 0's parameters have just replaced this rank's own draw inside the trainer: mark the variable dict (and
        the batch-norm-folded inference engine) stale, so that predict / score / export_weights on every rank see them."""
        if synced:
            self._dirty = True

    def _train_batch(self, batch):
        tr = self._ensure_trainer()
        x = self._to_device(batch[self.modality], torch.float32)
        labels = self._to_device(batch['labels'], torch.int32)
        self._graph = None          # a captured inference graph holds the pre-update weight pointers
        if self._reducer is not None and len(labels) != self.config['batchsize']:
            raise ValueError('data-parallel step with %d images, batchsize is %d' % (len(labels), self.config['batchsize']))
        self.loss = tr.step(x, labels, reducer=self._reducer)
        self._dirty = True
        return self.loss.item() if self.config.get('sync_loss', True) else 0.0

    def _sync_variables(self):
        """Master weights and moving statistics back into the variable dict; the inference engine folds every batch
        norm into its conv, so it is rebuilt from them."""
        if getattr(self, 'trainer', None) is not None and getattr(self, '_dirty', False):
            self.trainer.to_variables(self.variables)
            self._dirty = False
            self.engine.load(self.variables)

    def export_weights(self, save_dir=None):
        self._sync_variables()
        return BaseModel.export_weights(self, save_dir)

    def _predict_batch_impl(self, batch, output_attr=None):
        self._sync_variables()
        x = self._to_device(batch[self.modality], torch.float32)
        want = output_attr if output_attr in ('prob', 'score') else 'label'
        return self.engine.forward(x, want=(want,))[want]
