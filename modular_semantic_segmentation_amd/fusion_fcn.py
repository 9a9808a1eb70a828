"""fusion_fcn: the joint two-stream FCN baseline (reference: xview/models/fusion_fcn.py:11-40 on top of
xview/models/vgg16.py:7-51 and simple_fcn.decoder, simple_fcn.py:90-134) on the MI355X kernels.

One VGG16 trunk per modality (same conv kernels as the FCN experts, pools fused into the conv epilogues), channel
concat of the conv4_3 / conv5_3 maps, 1x1 `fused_score_conv4/5`, x2 bilinear + add, and the decoder head.  The
reference calls decoder() without `batchnorm`, i.e. with its default batch norm on `fused/upscore` and
`fused/score`: the first goes through the general (un-commuted) decoder head unless it is scale-only, the second
is folded into the 1x1 score weights.

Functional `vgg16` / `fusion_fcn` as used by experiments/timing.py:24-45, and the `FusionFCN` model with predict /
score / import / export and `fit` (trainer.FusionFcnTrainer: batch norm of the decoder in training mode).
"""
import numpy as np
import torch

from . import ops
from .base_model import BaseModel
from .custom_layers import bilinear_filter, dense_deconv_as_conv3x3, is_bilinear_filter
from .fcn import BN_EPS, ENCODER, _FUSE_FIRST, _fold_bn, padded_units


def vgg16_variable_shapes(prefix, in_channels):
    """vgg16.py:18-37: conv layers named '{prefix}_convX_Y' (no variable scope)."""
    shapes = {}
    cin = in_channels
    for name, cout, _ in ENCODER:
        shapes['%s_%s/kernel' % (prefix, name)] = (3, 3, cin, cout)
        shapes['%s_%s/bias' % (prefix, name)] = (cout,)
        cin = cout
    return shapes


def variable_shapes(prefixes, num_channels, num_units, num_classes, decoder_batch_norm=True):
    """name -> shape of every variable of fusion_fcn() (fusion_fcn.py:11-40)."""
    shapes = {}
    for m, prefix in prefixes.items():
        shapes.update(vgg16_variable_shapes(prefix, num_channels[m]))
    e = len(prefixes)
    for name in ('fused_score_conv4', 'fused_score_conv5'):
        shapes[name + '/kernel'] = (1, 1, 512 * e, num_units)
        shapes[name + '/bias'] = (num_units,)
    shapes['fused_upscore_conv5/kernel'] = (4, 4, num_units, num_units)
    shapes['fused/upscore/kernel'] = (16, 16, num_units, num_units)
    shapes['fused/score/kernel'] = (1, 1, num_units, num_classes)
    shapes['fused/score/bias'] = (num_classes,)
    if decoder_batch_norm:
        for layer, c in (('fused/upscore', num_units), ('fused/score', num_classes)):
            for v in ('gamma', 'beta', 'moving_mean', 'moving_variance'):
                shapes['%s/%s' % (layer, v)] = (c,)
    return shapes


def init_variables(prefixes, num_channels, num_units, num_classes, seed=None):
    """[TF1] default initialisers (Glorot-uniform kernels, zero biases, bilinear deconv constants, BN identity)."""
    rng = np.random.default_rng(seed)
    out = {}
    for name, shape in variable_shapes(prefixes, num_channels, num_units, num_classes).items():
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


class VggTrunk(object):
    """vgg16() for one modality: 13 convs, pool1..pool4 fused into the conv epilogues."""

    def __init__(self, prefix, in_channels, variables, device):
        self.prefix, self.cin, self.device = prefix, int(in_channels), torch.device(device)
        self._arena = {}
        self.load(variables)

    def load(self, variables):
        self.w, self.b = {}, {}
        for name, shape in vgg16_variable_shapes(self.prefix, self.cin).items():
            if name not in variables:
                raise KeyError('missing variable %s' % name)
            if tuple(np.shape(variables[name])) != tuple(shape):
                raise ValueError('variable %s has shape %s, expected %s' % (name, np.shape(variables[name]), shape))

        def up(a):
            return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(self.device)

        for name, _, _ in ENCODER:
            layer = '%s_%s' % (self.prefix, name)
            k, b = _fold_bn(variables, layer, np.asarray(variables[layer + '/kernel'], np.float32),
                            np.asarray(variables[layer + '/bias'], np.float32))
            self.w[name] = up(k) if name == 'conv1_1' else ops.pack_conv_weights(up(k))
            self.b[name] = up(b)

    def _act(self, name, n, h, w, c):
        key = (name, n, h, w, c)
        a = self._arena.get(key)
        if a is None:
            a = self._arena[key] = ops.Act(n, h, w, c, self.device)
        return a

    def forward(self, x, keep_all=False, routed=False):
        """x: float32 [N,H,W,cin] device tensor -> dict of Acts (always 'conv4_3', 'conv5_3').  As FcnEngine.encoder:
        without keep_all, conv1_1 + conv1_2 + pool1 are one launch (ops.conv_first_pair_fwd: neither full-resolution map is
        written); routed=True (with keep_all: the training step) leaves out the full map of a conv in front of a pool, whose
        only reader would be MaxPoolGrad, and keeps the pool's route bytes ('route_<conv>') for the data-gradient conv."""
        n, h, w, cin = x.shape
        if cin != self.cin:
            raise ValueError('expected %d input channels, got %d' % (self.cin, cin))
        if h % 16 or w % 16:
            raise ValueError('H and W must be multiples of 16')
        L = {}
        ch, cw = h, w
        first = 1
        if not keep_all and _FUSE_FIRST and ENCODER[1][0] == 'conv1_2' and ENCODER[1][2] == 'pool1':
            q = self._act('pool1', n, h // 2, w // 2, 64)
            if ops.conv_first_pair_fwd(x.contiguous(), self.w['conv1_1'], self.b['conv1_1'], self.w['conv1_2'], self.b['conv1_2'],
                                       pooled=q):
                L['pool1'] = cur = q
                ch, cw = h // 2, w // 2
                first = 2
        if first == 1:
            cur = self._act('conv1_1', n, h, w, 64)
            ops.conv2d_first_fwd(x.contiguous(), self.w['conv1_1'], self.b['conv1_1'], cur, relu=True)
            L['conv1_1'] = cur
        for name, cout, pool in ENCODER[first:]:
            if pool is None:
                y = self._act(name, n, ch, cw, cout)
                ops.conv2d_fwd(cur, self.w[name], self.b[name], 3, relu=True, y=y)
                L[name] = cur = y
            else:
                q = self._act(pool, n, ch // 2, cw // 2, cout)
                if routed and keep_all and name != 'conv4_3':
                    key = ('route_' + name, n, q.h, q.w, q.c)
                    route = self._arena.get(key)
                    if route is None:
                        route = self._arena[key] = torch.empty(n * q.h * q.w * q.c, dtype=torch.uint8, device=self.device)
                    if ops.conv2d_fwd_route(cur, self.w[name], self.b[name], q, route):
                        L['route_' + name] = route
                        L[pool] = cur = q
                        ch, cw = ch // 2, cw // 2
                        continue
                need_full = keep_all or name == 'conv4_3'
                y = self._act(name, n, ch, cw, cout) if need_full else None
                ops.conv2d_fwd(cur, self.w[name], self.b[name], 3, relu=True, y=y, pooled=q, write_y=need_full)
                if y is not None:
                    L[name] = y
                L[pool] = cur = q
                ch, cw = ch // 2, cw // 2
        return L


class FusionFcnEngine(object):
    """The whole fusion_fcn() inference graph resident on one GPU."""

    def __init__(self, prefixes, num_channels, num_units, num_classes, variables, device='cuda'):
        self.prefixes = dict(prefixes)
        self.num_channels = {m: int(num_channels[m]) for m in self.prefixes}
        self.U, self.C = int(num_units), int(num_classes)
        self.Up = padded_units(self.U)
        self.device = torch.device(device)
        self._arena = {}
        self.trunks = {}
        self.load(variables)

    def load(self, variables):
        v = {k: np.asarray(a, np.float32) for k, a in variables.items()}
        dev = self.device
        has_bn = 'fused/upscore/gamma' in v
        for need, shape in variable_shapes(self.prefixes, self.num_channels, self.U, self.C, has_bn).items():
            if need not in v:
                raise KeyError('missing variable %s' % need)
            if tuple(v[need].shape) != tuple(shape):
                raise ValueError('variable %s has shape %s, expected %s' % (need, v[need].shape, shape))
        # The reference never trains its deconvs (fusion_fcn.py:26-31: trainable=False), so their kernels are the bilinear
        # constant and run as depthwise interpolations.  An imported kernel that is anything else takes the dense
        # transposed-conv path, as in fcn.FcnEngine (xv_deconv_dense_fwd on the MFMA conv; the decoder head is then the
        # un-commuted one: dense x8 deconv -> [batch norm] -> relu, per-pixel score conv, softmax, argmax).
        self.dense_deconv = {}
        for name, stride in (('fused_upscore_conv5', 2), ('fused/upscore', 8)):
            kern = v[name + '/kernel']
            if not is_bilinear_filter(kern):
                kp = np.zeros((kern.shape[0], kern.shape[1], self.Up, self.Up), np.float32)
                kp[:, :, :self.U, :self.U] = kern                       # padding units: zero rows and columns
                k3 = torch.from_numpy(dense_deconv_as_conv3x3(kp, stride)).to(dev)
                self.dense_deconv[name] = (ops.pack_conv_weights(k3),
                                           torch.zeros(stride * stride * self.Up, dtype=torch.float32, device=dev))
        for m, prefix in self.prefixes.items():
            if m in self.trunks:
                self.trunks[m].load(v)
            else:
                self.trunks[m] = VggTrunk(prefix, self.num_channels[m], v, dev)

        def up(a):
            return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)

        self.w, self.b = {}, {}
        e = len(self.prefixes)
        for name in ('fused_score_conv4', 'fused_score_conv5'):
            k, b = _fold_bn(v, name, v[name + '/kernel'], v[name + '/bias'])
            kp = np.zeros((1, 1, 512 * e, self.Up), np.float32)
            kp[..., :self.U] = k
            bp = np.zeros(self.Up, np.float32)
            bp[:self.U] = b
            self.w[name] = ops.pack_conv_weights(up(kp))
            self.b[name] = up(bp)
        # decoder: BN on the 1x1 score conv (no activation) folds exactly; BN between the x8 deconv and its relu is
        # folded when scale-only, else handed to the general decoder head as a per-channel affine
        k, b = _fold_bn(v, 'fused/score', v['fused/score/kernel'], v['fused/score/bias'])
        k = k.reshape(self.U, self.C)
        self.head_affine = None
        if has_bn:
            s = v['fused/upscore/gamma'] / np.sqrt(v['fused/upscore/moving_variance'] + BN_EPS)
            t = v['fused/upscore/beta'] - v['fused/upscore/moving_mean'] * s
            if np.all(s > 0) and np.all(np.abs(t) <= 1e-12) and 'fused/upscore' not in self.dense_deconv:
                k = k * s[:, None]
            else:
                sp, tp = np.ones(self.Up, np.float32), np.zeros(self.Up, np.float32)
                sp[:self.U], tp[:self.U] = s, t
                self.head_affine = (up(sp), up(tp))
        ws = np.zeros((self.Up, self.C), np.float32)
        ws[:self.U] = k
        self.w['score'] = up(ws)
        self.b['score'] = up(b)
        torch.cuda.synchronize(dev)

    def _act(self, name, n, h, w, c):
        key = (name, n, h, w, c)
        a = self._arena.get(key)
        if a is None:
            a = self._arena[key] = ops.Act(n, h, w, c, self.device)
        return a

    def forward(self, inputs, want=('label',), keep_all=False):
        """inputs: {modality: float32 [N,H,W,c] device tensor}.  Returns dict with any of 'score', 'prob',
        'label' plus 'layers' (the reference's layer dict: per modality the trunk, then the fused layers)."""
        layers = {}
        mods = list(self.prefixes)
        # the trunks are independent up to the concat: one HIP stream each (as basic_fusion_model.run_experts does for
        # the experts), so that the tail round of one trunk's persistent conv grid is filled by the other's workgroups
        if not getattr(self, 'concurrent', True):
            for m in mods:
                layers[m] = self.trunks[m].forward(inputs[m], keep_all=keep_all)
        else:
            main = torch.cuda.current_stream(self.device)
            if not hasattr(self, '_streams'):
                self._streams = {m: torch.cuda.Stream(device=self.device) for m in mods}
            for m in mods:
                side = self._streams[m]
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    layers[m] = self.trunks[m].forward(inputs[m], keep_all=keep_all)
            for m in mods:
                main.wait_stream(self._streams[m])
        first = layers[mods[0]]['conv4_3']
        n, h8, w8 = first.n, first.h, first.w

        def concat(name):
            acts = [layers[m][name] for m in mods]
            cur = acts[0]
            for i, nxt in enumerate(acts[1:]):
                y = self._act('concat_%s_%d' % (name, i), cur.n, cur.h, cur.w, cur.c + nxt.c)
                ops.concat_channels(cur, nxt, y)
                cur = y
            return cur

        c4, c5 = concat('conv4_3'), concat('conv5_3')
        s4 = self._act('score_conv4', n, h8, w8, self.Up)
        ops.conv2d_fwd(c4, self.w['fused_score_conv4'], self.b['fused_score_conv4'], 1, relu=True, y=s4)
        s5 = self._act('score_conv5', n, h8 // 2, w8 // 2, self.Up)
        ops.conv2d_fwd(c5, self.w['fused_score_conv5'], self.b['fused_score_conv5'], 1, relu=True, y=s5)
        feat = self._act('features', n, h8, w8, self.Up)
        if 'fused_upscore_conv5' in self.dense_deconv:
            wk, zb = self.dense_deconv['fused_upscore_conv5']
            _, self._arena['dd_ws5'] = ops.deconv_dense_fwd(s5, wk, zb, 2, self.Up, y=feat, residual=s4, relu=True,
                                                           workspace=self._arena.get('dd_ws5'))
        else:
            ops.upsample2x_relu_add(s5, residual=s4, y=feat)
        aff = self.head_affine or (None, None)
        want_label = 'label' in want or 'classification' in want
        layers.update(concat_conv4=c4, concat_conv5=c5, score_conv4=s4, score_conv5=s5, features=feat)
        if 'fused/upscore' in self.dense_deconv:
            wk, zb = self.dense_deconv['fused/upscore']
            up = self._act('upscore', n, 8 * h8, 8 * w8, self.Up)
            _, self._arena['dd_ws'] = ops.deconv_dense_fwd(feat, wk, zb, 8, self.Up, y=up, scale=aff[0], shift=aff[1],
                                                          relu=True, workspace=self._arena.get('dd_ws'))
            skey = ('dense_score', n, h8, w8)
            if skey not in self._arena:
                self._arena[skey] = torch.empty((n, 8 * h8, 8 * w8, self.C), dtype=torch.float32, device=self.device)
            score = ops.score_dense_fwd(up, self.w['score'], self.b['score'], self.C, self._arena[skey])
            prob, label = ops.softmax_argmax(score, want_prob='prob' in want, want_label=want_label)
            out = {}
            layers['upscore'] = up
            if 'score' in want:
                out['score'] = score
            if prob is not None:
                out['prob'] = prob
            if label is not None:
                out['label'] = label
        else:
            out = ops.decoder_head_fwd(feat, self.w['score'], self.b['score'], self.C, want_score='score' in want,
                                       want_prob='prob' in want, want_label=want_label, scale=aff[0], shift=aff[1])
        out['layers'] = layers
        return out


# ---- functional entry points (experiments/timing.py:10,24-45) --------------------------------------------------
_TRUNKS, _ENGINES = {}, {}


def vgg16(inputs, prefix, params=None, variables=None):
    """Functional vgg16() (vgg16.py:7-51): dict of all layer outputs (ops.Act).  `params` (the reference's conv
    keyword dict) is accepted and ignored; variables: '{prefix}_convX_Y/kernel|bias' dict (default: [TF1]
    initialisers)."""
    cin = int(inputs.shape[-1])
    key = (prefix, cin, id(variables))
    ent = _TRUNKS.get(key)      # (variables dict, trunk): the entry keeps the dict alive -- its id is part of the key -- and checks identity
    t = ent[1] if ent is not None and ent[0] is variables else None
    given = variables
    if t is None:
        if variables is None:
            rng = np.random.default_rng(0)
            variables = {}
            for name, shape in vgg16_variable_shapes(prefix, cin).items():
                if name.endswith('/kernel'):
                    lim = np.sqrt(6.0 / (9 * shape[2] + 9 * shape[3]))
                    variables[name] = rng.uniform(-lim, lim, size=shape).astype(np.float32)
                else:
                    variables[name] = np.zeros(shape, np.float32)
        t = VggTrunk(prefix, cin, variables, inputs.device)
        _TRUNKS[key] = (given, t)
    return t.forward(inputs, keep_all=True)


def fusion_fcn(inputs, prefixes, num_units, num_classes, variables=None, **unused):
    """Functional fusion_fcn() (fusion_fcn.py:11-40): the reference's layer dict with 'score' (float32
    [N,H,W,C]) plus 'prob' / 'classification' from the same head kernel.  trainable / is_training / reuse are
    accepted and ignored (inference graph)."""
    num_channels = {m: int(inputs[m].shape[-1]) for m in prefixes}
    key = (tuple(sorted(prefixes.items())), tuple(sorted(num_channels.items())), int(num_units), int(num_classes),
           id(variables))
    ent = _ENGINES.get(key)
    eng = ent[1] if ent is not None and ent[0] is variables else None
    if eng is None:
        given = variables
        if variables is None:
            variables = init_variables(prefixes, num_channels, num_units, num_classes)
        eng = FusionFcnEngine(prefixes, num_channels, num_units, num_classes, variables,
                              device=next(iter(inputs.values())).device)
        _ENGINES[key] = (given, eng)
    out = eng.forward(inputs, want=('score', 'prob', 'label'), keep_all=True)
    layers = dict(out['layers'])
    layers.update(score=out['score'], prob=out['prob'], classification=out['label'])
    return layers


class FusionFCN(BaseModel):
    """Model class of the joint baseline.  The reference's class (fusion_fcn.py:43-50) predates its current
    BaseModel signature; the arguments here are its own -- prefixes, num_channels, num_units, num_classes,
    trainer, learning_rate, output_dir -- and the data description is derived from them."""

    def __init__(self, prefixes, num_channels, num_units, num_classes, trainer='rmsprop', learning_rate=0.0001,
                 output_dir=None, **config):
        self.modalities = list(prefixes.keys())
        dtypes = {m: 'float32' for m in self.modalities}
        shapes = {m: (None, None, int(num_channels[m])) for m in self.modalities}
        dtypes['labels'], shapes['labels'] = 'int32', (None, None)
        BaseModel.__init__(self, (dtypes, shapes, int(num_classes)), name='FusionFCN', output_dir=output_dir,
                           prefixes=dict(prefixes), num_channels=dict(num_channels), num_units=num_units,
                           trainer=trainer, learning_rate=learning_rate, **config)

    def _build_graph(self):
        cfg = self.config
        self.variables = init_variables(cfg['prefixes'], cfg['num_channels'], cfg['num_units'], cfg['num_classes'],
                                        seed=cfg.get('seed'))
        self.engine = FusionFcnEngine(cfg['prefixes'], cfg['num_channels'], cfg['num_units'], cfg['num_classes'],
                                      self.variables, device=self.device)
        self.loss = None
        self.prediction = 'label'

    def _variables_changed(self):
        BaseModel._variables_changed(self)
        self.engine.load(self.variables)
        if getattr(self, 'trainer', None) is not None:
            from .parallel import sync_trainer_from_rank0
            self.trainer.load_from_variables(self.variables)
            self._after_rank0_sync(sync_trainer_from_rank0(self.trainer))

    # ---- training (FusionFCN._build_graph :50-92; optimizer setup base_model.py:153-162) -------------------------
    def _ensure_trainer(self):
        if getattr(self, 'trainer', None) is None:
            from .trainer import FusionFcnTrainer
            from .parallel import GradReducer, require_equal_batchsize, sync_trainer_from_rank0, world
            self.trainer = FusionFcnTrainer(self.engine, self.config.get('trainer', 'rmsprop'),
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
        x = {m: self._to_device(batch[m], torch.float32) for m in self.modalities}
        labels = self._to_device(batch['labels'], torch.int32)
        self._graph = None          # a captured inference graph holds the pre-update weight pointers
        if self._reducer is not None and len(labels) != self.config['batchsize']:
            raise ValueError('data-parallel step with %d images, batchsize is %d' % (len(labels), self.config['batchsize']))
        self.loss = tr.step(x, labels, reducer=self._reducer)
        self._dirty = True
        return self.loss.item() if self.config.get('sync_loss', True) else 0.0

    def _sync_variables(self):
        """Master weights and moving statistics back into the variable dict; the inference engine folds the batch norm
        of the decoder into its weights, so it is rebuilt from them."""
        if getattr(self, 'trainer', None) is not None and getattr(self, '_dirty', False):
            self.trainer.to_variables(self.variables)
            self._dirty = False
            self.engine.load(self.variables)

    def export_weights(self, save_dir=None):
        self._sync_variables()
        return BaseModel.export_weights(self, save_dir)

    def _predict_batch_impl(self, batch, output_attr=None):
        self._sync_variables()
        x = {m: self._to_device(batch[m], torch.float32) for m in self.modalities}
        self.engine.concurrent = getattr(self, 'concurrent_experts', True)     # one stream per trunk unless told otherwise
        want = output_attr if output_attr in ('prob', 'score') else 'label'
        return self.engine.forward(x, want=(want,))[want]
