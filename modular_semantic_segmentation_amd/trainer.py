"""Training step of one FCN expert on MI355X: forward with saved activations, backward through the
HIP kernels, [TF1]-semantics optimizer, data-parallel gradient all-reduce.

Mirrors what `sess.run(self.trainer)` executes in the reference (base_model.py:153-162,257-258 over
simple_fcn.py:200-214): loss = cross_entropy(log_softmax(fcn(x)), one_hot(labels)), then
tf.train.{Adam,RMSProp,Adagrad}Optimizer(learning_rate).minimize(loss).

Parameters live in ONE flat float32 buffer (master weights) laid out in BACKWARD order, so that the
gradient buffer splits into contiguous buckets that become ready one after the other during the
backward pass; each bucket is all-reduced (RCCL over xGMI) on a side HIP stream while the remaining
layers are still differentiating.
"""
import collections
import contextlib
import os

import numpy as np
import torch

from . import ops
from .fcn import ENCODER

# XV_VIRTUAL_UPSCORE=0: the batch-norm trainers store the x8 deconv's output and read it back (A/B timing)
_VIRTUAL_UPSCORE = os.environ.get('XV_VIRTUAL_UPSCORE', '1') != '0'
# filter gradients on a second HIP stream (encoder_backward); XV_WGRAD_STREAM=0: everything on one stream (A/B timing)
_WGRAD_STREAM = os.environ.get('XV_WGRAD_STREAM', '1') != '0'


def _ups8_channels_ok(c):
    """Channel counts the batch-norm kernels take (batchnorm.hip: c >= 64 and 2048 % c == 0 -- 64, 128, 256, ...; the
    xv_bn_*_ups8 entry points, ups_ok, the same).  fcn.padded_units pads num_units <= 256 to one of them (129..192 used to be
    padded to 192 lanes, for which no statistics kernel exists: batch-norm training failed with XV_ESHAPE)."""
    return c >= 64 and 2048 % c == 0

# backward order of the trainable layers
LAYER_ORDER = ['score', 'score_conv5', 'score_conv4'] + [name for name, _, _ in reversed(ENCODER)]
BUCKETS = [['score', 'score_conv5', 'score_conv4', 'conv5_3', 'conv5_2', 'conv5_1'],
           ['conv4_3', 'conv4_2', 'conv4_1'],
           ['conv3_3', 'conv3_2', 'conv3_1', 'conv2_2', 'conv2_1', 'conv1_2', 'conv1_1']]


_Shape = collections.namedtuple('_Shape', 'n h w c')      # what gact() reads of its `like` argument


def encoder_backward(x, L, g, ds4, wd_s4, wd, zero_bias, G, gact, wws, after_layer=None, wstream=None):
    """Backward walk over the 13 trunk convs (relu + fused pools) of one modality.
    x: raw network input; L: forward layer dict (every convX_Y and poolX); g: gradient w.r.t. conv5_3's output, already
    masked by its relu; ds4 / wd_s4: gradient of the 1x1 score conv on conv4_3 and its packed data-gradient weights
    (the second path into conv4_3, AddN); wd[name]: packed data-gradient weights; G(name, kind): gradient views;
    gact(like, tag): scratch activations; after_layer(name): called once a layer's filter gradient is complete.
    wstream: a second HIP stream for the filter gradients.  Nothing in the walk reads them (the optimizer does, after the
    join at the end), every layer's gradient map and activation has a buffer of its own, and the one slab workspace is
    used by this stream alone -- so a layer's filter gradient runs beside the next layers' data gradients, and its
    workgroups fill the half-empty last rounds of the persistent data-gradient grids (conv4_x: 4.5 rounds, conv5_x: 1.5)."""
    names = [nm for nm, _, _ in ENCODER]
    pool_after = {nm: pl for nm, _, pl in ENCODER}
    inputs, prev = {}, None
    for nm in names:
        inputs[nm] = prev
        prev = pool_after[nm] if pool_after[nm] else nm
    main = torch.cuda.current_stream(x.device) if wstream is not None else None

    def filter_gradient(nm, xin, g):
        if nm == 'conv1_1':
            ops.conv2d_first_bwd_filter(x, g, G(nm, 'kernel'), G(nm, 'bias'), workspace=wws)
        else:
            ops.conv2d_bwd_filter(L[xin], g, G(nm, 'kernel'), G(nm, 'bias'), 3, workspace=wws)
        if after_layer is not None:
            after_layer(nm)

    for nm in reversed(names):
        xin = inputs[nm]
        if wstream is None:
            filter_gradient(nm, xin, g)
        else:
            ready = torch.cuda.Event()
            ready.record(main)                   # g (and, the first time, the zeroed gradient buffer) is complete
            with torch.cuda.stream(wstream):
                wstream.wait_event(ready)
                filter_gradient(nm, xin, g)
        if nm == 'conv1_1':
            if wstream is not None:
                main.wait_stream(wstream)
            break
        if xin.startswith('pool'):
            # gradient w.r.t. the pooled map, then MaxPoolGrad + ReluGrad onto the conv above
            above = names[names.index(nm) - 1]
            route = L.get('route_' + above)
            if route is not None:
                # ... in the data-gradient conv's own epilogue, through the route bytes the forward conv left (the pooled
                # gradient never reaches memory; the full map `above` was never written)
                p = L[xin]
                g = ops.conv2d_bwd_data_route(g, wd[nm], zero_bias, route, gact(_Shape(p.n, 2 * p.h, 2 * p.w, p.c), 'r_' + above))
                continue
            dpool = ops.conv2d_bwd_data(g, wd[nm], zero_bias, gact(L[xin], 'g_' + xin), 3)
            routed = ops.maxpool2x2_bwd(L[above], dpool, gact(L[above], 'r_' + above))
            if above == 'conv4_3':
                # second gradient path into conv4_3: through the 1x1 score conv (AddN), then its relu
                g = ops.conv2d_bwd_data(ds4, wd_s4, zero_bias, gact(L[above], 'g_' + above), 1, relu_ref=L[above],
                                        addend=routed)
            else:
                g = routed
        else:
            g = ops.conv2d_bwd_data(g, wd[nm], zero_bias, gact(L[xin], 'g_' + xin), 3, relu_ref=L[xin])


def init_optimizer_state(tr):
    """[TF1] slot initial values: Adam m = v = 0; RMSProp ms = 1 (momentum 0); Adagrad accumulator = 0.1."""
    if tr.kind == 'adam':
        tr.state = {'m': torch.zeros_like(tr.param), 'v': torch.zeros_like(tr.param)}
    elif tr.kind == 'rmsprop':
        tr.state = {'ms': torch.ones_like(tr.param)}
    else:
        tr.state = {'acc': torch.full_like(tr.param, 0.1)}


class FcnTrainer(object):
    def __init__(self, engine, trainer='adam', learning_rate=1e-4):
        self.e = engine
        self.kind = trainer
        self.lr = float(learning_rate)
        if trainer not in ('adam', 'rmsprop', 'adagrad'):
            raise KeyError(trainer)
        e = engine
        dev = e.device
        # ---- flat parameter layout (padded U for the score layers) ---------------------------------
        self.shapes = {}
        cin = e.cin
        for name, cout, _ in ENCODER:
            self.shapes[name] = ((3, 3, cin, cout), (cout,))
            cin = cout
        self.shapes['score_conv4'] = ((1, 1, 512, e.Up), (e.Up,))
        self.shapes['score_conv5'] = ((1, 1, 512, e.Up), (e.Up,))
        self.shapes['score'] = ((e.Up, e.C), (e.C,))
        self.offsets, total = {}, 0
        self.bucket_ranges = []
        for bucket in BUCKETS:
            b0 = total
            for name in bucket:
                for kind, shape in zip(('kernel', 'bias'), self.shapes[name]):
                    n = int(np.prod(shape))
                    n_al = (n + 63) // 64 * 64           # keep every tensor 256-byte aligned
                    self.offsets[(name, kind)] = (total, n, shape)
                    total += n_al
            self.bucket_ranges.append((b0, total))
        self.total = total
        self.param = torch.zeros(total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.state = {}
        self.t = 0
        self.zero_bias = torch.zeros(512, dtype=torch.float32, device=dev)
        self.count = torch.zeros(1, dtype=torch.int64, device=dev)
        self.loss = torch.zeros(1, dtype=torch.float64, device=dev)
        self.wd = {}                                   # packed dgrad weights per layer
        self._g = {}
        self._side = None
        self.load_from_variables(e_variables=None)

    # ---- views ------------------------------------------------------------------------------------------
    def view(self, buf, name, kind):
        off, n, shape = self.offsets[(name, kind)]
        return buf[off:off + n].view(*shape)

    def load_from_variables(self, e_variables=None, variables=None):
        """Fill the master buffer from a reference-schema variable dict and (re)pack the kernels."""
        if variables is not None:
            p = self.e.prefix
            for name in LAYER_ORDER:
                k = np.asarray(variables['%s/%s/kernel' % (p, name)], np.float32)
                b = np.asarray(variables['%s/%s/bias' % (p, name)], np.float32)
                kv, bv = self.view(self.param, name, 'kernel'), self.view(self.param, name, 'bias')
                kv.zero_()
                bv.zero_()
                if name == 'score':
                    kv[:self.e.U].copy_(torch.from_numpy(k.reshape(self.e.U, self.e.C)))
                    bv.copy_(torch.from_numpy(b))
                elif name.startswith('score_conv'):
                    kv[..., :self.e.U].copy_(torch.from_numpy(k))
                    bv[:self.e.U].copy_(torch.from_numpy(b))
                else:
                    kv.copy_(torch.from_numpy(k))
                    bv.copy_(torch.from_numpy(b))
            self.repack()

    def to_variables(self, variables):
        """Write the master weights back into a reference-schema dict of numpy arrays."""
        p = self.e.prefix
        for name in LAYER_ORDER:
            kv = self.view(self.param, name, 'kernel').cpu().numpy()
            bv = self.view(self.param, name, 'bias').cpu().numpy()
            if name == 'score':
                kv, bv = kv[:self.e.U].reshape(1, 1, self.e.U, self.e.C), bv
            elif name.startswith('score_conv'):
                kv, bv = kv[..., :self.e.U], bv[:self.e.U]
            variables['%s/%s/kernel' % (p, name)] = np.ascontiguousarray(kv)
            variables['%s/%s/bias' % (p, name)] = np.ascontiguousarray(bv)

    def repack(self):
        """Master fp32 weights -> the engine's bf16 packed forward weights + packed dgrad weights: ONE launch over a
        descriptor table (built once: the master views and the packed buffers never move).  The engine is pointed at
        these buffers every time (FcnEngine.load() installs buffers of its own when variables are imported)."""
        e = self.e
        if getattr(self, '_pack_table', None) is None:
            entries, self._fwd = [], {}
            for name in LAYER_ORDER:
                kv = self.view(self.param, name, 'kernel')
                if name in ('score', 'conv1_1'):
                    self._fwd[name] = kv
                    continue
                k, _, cin, cout = kv.shape
                nel = ops.packed_weight_elems(k, cin, cout)
                self._fwd[name] = torch.empty(nel, dtype=torch.bfloat16, device=e.device)
                self.wd[name] = torch.empty(nel, dtype=torch.bfloat16, device=e.device)
                entries.append((kv, self._fwd[name], self.wd[name]))
            self._pack_table = ops.PackTable(entries, e.device)
        for name in LAYER_ORDER:
            e.w[name] = self._fwd[name]
            e.b[name] = self.view(self.param, name, 'bias')
        self._pack_table.run()

    @staticmethod
    def _input_of(name):
        """Layer whose output feeds conv `name` (a conv or a pool)."""
        prev = None
        for nm, _, pool in ENCODER:
            if nm == name:
                return prev
            prev = pool if pool else nm
        raise KeyError(name)

    def _gact(self, like, tag):
        key = (tag, like.n, like.h, like.w, like.c)
        a = self._g.get(key)
        if a is None:
            a = ops.Act(like.n, like.h, like.w, like.c, self.e.device)
            self._g[key] = a
        return a

    # ---- one training step ----------------------------------------------------------------------------------
    def step(self, x, labels, reducer=None):
        """x: float32 [N,H,W,cin], labels: int32 [N,H,W] (device tensors).  Returns the loss (device
        float64 scalar tensor).  reducer: parallel.GradReducer for data-parallel runs."""
        e = self.e
        L = e.encoder(x, keep_all=True, routed=True)
        n, h, w, _ = x.shape
        ops.zero_(self.grad)                            # (the library's memset: no framework kernel runs in the step)
        ops.zero_(self.loss)
        ops.zero_(self.count)
        ops.count_valid_labels(labels, e.C, self.count)
        if reducer is not None:
            reducer.allreduce_now(self.count)           # loss denominator = labelled pixels of the GLOBAL batch
        G = lambda name, kind: self.view(self.grad, name, kind)   # noqa: E731
        # one workspace for the split-K slabs of every filter gradient (largest layer decides)
        wkey = ('wgrad_ws', n, h, w)
        if wkey not in self._g:
            need = 0
            for nm in [m for m, _, _ in ENCODER[1:]]:
                need = max(need, ops.conv2d_bwd_filter_workspace_bytes(L[self._input_of(nm)], self.shapes[nm][1][0], 3))
            for nm, src in (('score_conv4', 'conv4_3'), ('score_conv5', 'conv5_3')):
                need = max(need, ops.conv2d_bwd_filter_workspace_bytes(L[src], e.Up, 1))
            need = max(need, ops.conv2d_first_bwd_filter_workspace_bytes(x))
            self._g[wkey] = torch.empty(need // 4, dtype=torch.float32, device=e.device)
        wws = self._g[wkey]
        dfused = self._gact(L['fused'], 'dfused')
        key = ('head_ws', n, h, w)
        self._g[key] = ops.decoder_head_bwd(L['fused'], e.w['score'], e.b['score'], labels, self.count, e.C, self.loss,
                                            G('score', 'kernel'), G('score', 'bias'), dfused,
                                            workspace=self._g.get(key))
        ds4 = ops.relu_bwd(dfused, L['score_conv4'], self._gact(L['score_conv4'], 'ds4'))
        ds5 = ops.upsample2x_bwd(dfused, L['score_conv5'], self._gact(L['score_conv5'], 'ds5'))
        ops.conv2d_bwd_filter(L['conv5_3'], ds5, G('score_conv5', 'kernel'), G('score_conv5', 'bias'), 1, workspace=wws)
        ops.conv2d_bwd_filter(L['conv4_3'], ds4, G('score_conv4', 'kernel'), G('score_conv4', 'bias'), 1, workspace=wws)
        g = ops.conv2d_bwd_data(ds5, self.wd['score_conv5'], self.zero_bias, self._gact(L['conv5_3'], 'g_conv5_3'), 1,
                                relu_ref=L['conv5_3'])
        state = {'done': 0}

        def after_layer(nm):
            if reducer is not None and state['done'] < len(BUCKETS) and nm == BUCKETS[state['done']][-1]:
                reducer.launch(self.grad, self.bucket_ranges[state['done']])
                state['done'] += 1

        if _WGRAD_STREAM and x.is_cuda and getattr(self, '_wstream', None) is None:
            self._wstream = torch.cuda.Stream(device=x.device)
        encoder_backward(x, L, g, ds4, self.wd['score_conv4'], self.wd, self.zero_bias, G, self._gact, wws, after_layer,
                         wstream=self._wstream if _WGRAD_STREAM and x.is_cuda else None)
        scale = 1.0
        if reducer is not None:
            reducer.wait()
            scale = 1.0                                 # gradients are sums over the global batch / global denominator
        self.t += 1
        self._apply(scale)
        self.repack()
        return self.loss

    def _apply(self, scale):
        if not self.state:
            init_optimizer_state(self)
        if self.kind == 'adam':
            b1, b2 = 0.9, 0.999
            lr_t = self.lr * np.sqrt(1 - b2 ** self.t) / (1 - b1 ** self.t)
            ops.adam_step(self.param, self.grad, self.state['m'], self.state['v'], float(lr_t), b1, b2, 1e-8, scale)
        elif self.kind == 'rmsprop':
            ops.rmsprop_step(self.param, self.grad, self.state['ms'], self.lr, 0.9, 1e-10, scale)
        else:
            ops.adagrad_step(self.param, self.grad, self.state['acc'], self.lr, scale)

    def grads_as_variables(self):
        """Gradient of the last step in the reference schema (tests)."""
        out = {}
        p = self.e.prefix
        for name in LAYER_ORDER:
            kv = self.view(self.grad, name, 'kernel').cpu().numpy()
            bv = self.view(self.grad, name, 'bias').cpu().numpy()
            if name == 'score':
                kv = kv[:self.e.U].reshape(1, 1, self.e.U, self.e.C)
            elif name.startswith('score_conv'):
                kv, bv = kv[..., :self.e.U], bv[:self.e.U]
            out['%s/%s/kernel' % (p, name)] = kv
            out['%s/%s/bias' % (p, name)] = bv
        return out


# =========================================================================================================
# Training with batch normalisation (`batch_normalization: true`, the reference's example configuration)
# =========================================================================================================
BN_LAYERS = [name for name, _, _ in ENCODER] + ['score_conv4', 'score_conv5', 'upscore_conv5', 'upscore', 'score']
BN_ORDER = ['score', 'upscore', 'score_conv5', 'upscore_conv5', 'score_conv4'] + [name for name, _, _ in reversed(ENCODER)]
# gradient buckets of the data-parallel step, contiguous runs of BN_ORDER (every layer's kernel, bias, gamma and beta lie
# together): a bucket is complete once the filter gradient of its LAST layer has been enqueued (FcnTrainer's BUCKETS with the
# head's batch norms in the first one)
BN_BUCKETS = [BN_ORDER[:5] + ['conv5_3', 'conv5_2', 'conv5_1'], ['conv4_3', 'conv4_2', 'conv4_1'],
              ['conv3_3', 'conv3_2', 'conv3_1', 'conv2_2', 'conv2_1', 'conv1_2', 'conv1_1']]
assert [n for b in BN_BUCKETS for n in b] == BN_ORDER


class FcnBnTrainer(object):
    """One training step of SimpleFCN with tf.layers.batch_normalization(training=True) between every conv /
    deconv and its activation (custom_layers.py:112-119,124-139; simple_fcn.py:200-214).

    Differences from FcnTrainer, all forced by the batch statistics: convs run without their fused relu / pool and
    store the pre-normalisation map z next to the post-relu map y; the relu mask of the backward pass moves into the
    batch-norm gradient; the decoder head is un-commuted (the x8 bilinear map is materialised at full resolution,
    normalised, scored per pixel and normalised again).  gamma / beta are trained ([TF1] defaults), the moving
    averages (momentum 0.99, unbiased variance) are updated in the forward pass like the reference's UPDATE_OPS
    control dependency (base_model.py:155-156).  Data-parallel runs synchronise the batch statistics (Sync-BN)."""

    def __init__(self, engine, trainer='adam', learning_rate=1e-4):
        if trainer not in ('adam', 'rmsprop', 'adagrad'):
            raise KeyError(trainer)
        self.e, self.kind, self.lr = engine, trainer, float(learning_rate)
        e, dev = engine, engine.device
        if not _ups8_channels_ok(e.Up):
            raise ValueError('batch-norm training: num_units = %d is padded to %d lanes; the batch-norm kernels take 64, 128 '
                             'or 256 (num_units <= 256)' % (e.U, e.Up))
        self.convs = {}                      # conv layers: (kernel shape, cout)
        cin = e.cin
        for name, cout, _ in ENCODER:
            self.convs[name] = ((3, 3, cin, cout), cout)
            cin = cout
        self.convs['score_conv4'] = ((1, 1, 512, e.Up), e.Up)
        self.convs['score_conv5'] = ((1, 1, 512, e.Up), e.Up)
        self.convs['score'] = ((e.Up, e.C), e.C)
        self.bn_channels = {name: self.convs[name][1] for name in self.convs}
        self.bn_channels.update(upscore_conv5=e.Up, upscore=e.Up)
        self.offsets, total = {}, 0
        self.bucket_ranges = []
        for bucket in BN_BUCKETS:
            b0 = total
            for name in bucket:
                entries = []
                if name in self.convs:
                    entries += [('kernel', self.convs[name][0]), ('bias', (self.convs[name][1],))]
                entries += [('gamma', (self.bn_channels[name],)), ('beta', (self.bn_channels[name],))]
                for kind, shape in entries:
                    n = int(np.prod(shape))
                    self.offsets[(name, kind)] = (total, n, shape)
                    total += (n + 63) // 64 * 64
            self.bucket_ranges.append((b0, total))
        self.total = total
        self.param = torch.zeros(total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.moving = {name: (torch.zeros(c, device=dev), torch.ones(c, device=dev)) for name, c in self.bn_channels.items()}
        self.bn = {name: ops.BnState(c, dev) for name, c in self.bn_channels.items()}
        self.state, self.t = {}, 0
        self.zero_bias = torch.zeros(512, dtype=torch.float32, device=dev)
        self.keep_all = False   # True: also write the full-resolution activations of the pooled layers (tests, inspection)
        self.count = torch.zeros(1, dtype=torch.int64, device=dev)
        self.loss = torch.zeros(1, dtype=torch.float64, device=dev)
        self.w, self.wd, self._a = {}, {}, {}

    # ---- parameters -------------------------------------------------------------------------------------
    def view(self, buf, name, kind):
        off, n, shape = self.offsets[(name, kind)]
        return buf[off:off + n].view(*shape)

    def _real_channels(self, name):
        """Channels of a batch norm that exist in the reference graph (the U-channel layers are padded to Up lanes)."""
        if name == 'score':
            return self.e.C
        if name in ('score_conv4', 'score_conv5', 'upscore_conv5', 'upscore'):
            return self.e.U
        return self.bn_channels[name]

    def load_from_variables(self, e_variables=None, variables=None):
        if variables is None:
            return
        p, e = self.e.prefix, self.e
        self.param.zero_()
        for name in BN_ORDER:
            layer = '%s/%s' % (p, name)
            c = self.bn_channels[name]
            real = self._real_channels(name)
            if name in self.convs:
                k = torch.from_numpy(np.asarray(variables[layer + '/kernel'], np.float32))
                b = torch.from_numpy(np.asarray(variables[layer + '/bias'], np.float32))
                kv, bv = self.view(self.param, name, 'kernel'), self.view(self.param, name, 'bias')
                if name == 'score':
                    kv[:e.U].copy_(k.reshape(e.U, e.C))
                    bv.copy_(b)
                elif name.startswith('score_conv'):
                    kv[..., :e.U].copy_(k)
                    bv[:e.U].copy_(b)
                else:
                    kv.copy_(k)
                    bv.copy_(b)
            g, bt = self.view(self.param, name, 'gamma'), self.view(self.param, name, 'beta')
            g.fill_(1.0)                       # padding channels: gamma 1, beta 0 (their maps stay exactly zero)
            g[:real].copy_(torch.from_numpy(np.asarray(variables[layer + '/gamma'], np.float32)))
            bt[:real].copy_(torch.from_numpy(np.asarray(variables[layer + '/beta'], np.float32)))
            mm, mv = self.moving[name]
            mm.zero_()
            mv.fill_(1.0)
            mm[:real].copy_(torch.from_numpy(np.asarray(variables[layer + '/moving_mean'], np.float32)))
            mv[:real].copy_(torch.from_numpy(np.asarray(variables[layer + '/moving_variance'], np.float32)))
        self.repack()

    def to_variables(self, variables):
        p, e = self.e.prefix, self.e
        for name in BN_ORDER:
            layer = '%s/%s' % (p, name)
            c = self.bn_channels[name]
            real = self._real_channels(name)
            if name in self.convs:
                kv = self.view(self.param, name, 'kernel').cpu().numpy()
                bv = self.view(self.param, name, 'bias').cpu().numpy()
                if name == 'score':
                    kv = kv[:e.U].reshape(1, 1, e.U, e.C)
                elif name.startswith('score_conv'):
                    kv, bv = kv[..., :e.U], bv[:e.U]
                variables[layer + '/kernel'] = np.ascontiguousarray(kv)
                variables[layer + '/bias'] = np.ascontiguousarray(bv)
            variables[layer + '/gamma'] = self.view(self.param, name, 'gamma')[:real].cpu().numpy().copy()
            variables[layer + '/beta'] = self.view(self.param, name, 'beta')[:real].cpu().numpy().copy()
            variables[layer + '/moving_mean'] = self.moving[name][0][:real].cpu().numpy().copy()
            variables[layer + '/moving_variance'] = self.moving[name][1][:real].cpu().numpy().copy()

    def grads_as_variables(self):
        out = {}
        p, e = self.e.prefix, self.e
        for name in BN_ORDER:
            layer = '%s/%s' % (p, name)
            c = self.bn_channels[name]
            real = self._real_channels(name)
            if name in self.convs:
                kv = self.view(self.grad, name, 'kernel').cpu().numpy()
                bv = self.view(self.grad, name, 'bias').cpu().numpy()
                if name == 'score':
                    kv = kv[:e.U].reshape(1, 1, e.U, e.C)
                elif name.startswith('score_conv'):
                    kv, bv = kv[..., :e.U], bv[:e.U]
                out[layer + '/kernel'], out[layer + '/bias'] = kv, bv
            out[layer + '/gamma'] = self.view(self.grad, name, 'gamma')[:real].cpu().numpy()
            out[layer + '/beta'] = self.view(self.grad, name, 'beta')[:real].cpu().numpy()
        return out

    def repack(self):
        """Master fp32 kernels -> packed bf16 forward / data-gradient weights (training graph: no folding)."""
        dev = self.e.device
        if getattr(self, '_pack_table', None) is None:
            entries = []
            for name, (shape, _) in self.convs.items():
                kv = self.view(self.param, name, 'kernel')
                if name in ('score', 'conv1_1'):
                    self.w[name] = kv
                    continue
                k, _, cin, cout = shape
                nel = ops.packed_weight_elems(k, cin, cout)
                self.w[name] = torch.empty(nel, dtype=torch.bfloat16, device=dev)
                self.wd[name] = torch.empty(nel, dtype=torch.bfloat16, device=dev)
                entries.append((kv, self.w[name], self.wd[name]))
            self._pack_table = ops.PackTable(entries, dev)      # all kernels in ONE launch (the master views never move)
        self._pack_table.run()

    def _act(self, tag, n, h, w, c):
        key = (tag, n, h, w, c)
        a = self._a.get(key)
        if a is None:
            a = self._a[key] = ops.Act(n, h, w, c, self.e.device)
        return a

    def _dense(self, tag, shape, dtype=torch.float32):
        key = (tag,) + tuple(shape)
        t = self._a.get(key)
        if t is None:
            t = self._a[key] = torch.empty(shape, dtype=dtype, device=self.e.device)
        return t

    def _bn_fwd(self, name, z, y, relu=True, pooled=None, have_stats=False, ups8_of=None):
        mm, mv = self.moving[name]
        return ops.bn_forward(z, self.view(self.param, name, 'gamma'), self.view(self.param, name, 'beta'), mm, mv,
                              self.bn[name], y, relu=relu, sync=self._sync, pooled=pooled, have_stats=have_stats,
                              ups8_of=ups8_of)

    def _bn_bwd(self, name, dy, y, z, dz, ups8_of=None):
        return ops.bn_backward(dy, y, z, self.view(self.param, name, 'gamma'), self.bn[name],
                               self.view(self.grad, name, 'gamma'), self.view(self.grad, name, 'beta'), dz,
                               sync=self._sync, ups8_of=ups8_of)

    # ---- one training step ----------------------------------------------------------------------------------
    def step(self, x, labels, reducer=None):
        e = self.e
        n, h, w, _ = x.shape
        if h % 16 or w % 16:
            raise ValueError('H and W must be multiples of 16')
        P = lambda name, kind: self.view(self.param, name, kind)   # noqa: E731
        G = lambda name, kind: self.view(self.grad, name, kind)    # noqa: E731
        ops.zero_(self.grad)                            # (the library's memset: no framework kernel runs in the step)
        ops.zero_(self.loss)
        ops.zero_(self.count)
        ops.count_valid_labels(labels, e.C, self.count)
        # Sync-BN under data parallelism: batch statistics (and their gradient sums) over the GLOBAL batch, one small
        # all-reduce per batch norm and direction, so that N ranks reproduce a single device on the whole batch
        self._sync = reducer is not None
        if reducer is not None:
            reducer.allreduce_now(self.count)
        # ---- forward: z = conv + bias, y = relu(BN(z)), pools on y ---------------------------------------
        Z, Y = {}, {}
        cur, ch, cw = None, h, w
        inputs = {}
        for name, cout, pool in ENCODER:
            z = self._act('z_' + name, n, ch, cw, cout)
            have_stats = False
            if name == 'conv1_1':
                ops.conv2d_first_fwd(x.contiguous(), self.w[name], P(name, 'bias'), z, relu=False)
            else:
                # the batch statistics in the conv's own epilogue where its kernel allows (generation 4: exact 16x32 tilings)
                have_stats = ops.conv2d_fwd_stats(cur, self.w[name], P(name, 'bias'), z, self.bn[name])
                if not have_stats:
                    ops.conv2d_fwd(cur, self.w[name], P(name, 'bias'), 3, relu=False, y=z)
            inputs[name] = cur
            Z[name] = z
            if pool:
                # batch norm + relu + pool in one pass; the full-resolution activation is written only where something
                # reads it (conv4_3 feeds score_conv4; keep_all: inspection) -- the gradient passes recompute it from z
                ya = self._act('y_' + name, n, ch, cw, cout) if (name == 'conv4_3' or self.keep_all) else None
                ch, cw = ch // 2, cw // 2
                Y[pool] = cur = self._act(pool, n, ch, cw, cout)
                Y[name] = self._bn_fwd(name, z, ya, pooled=cur, have_stats=have_stats)
            else:
                Y[name] = cur = self._bn_fwd(name, z, self._act('y_' + name, n, ch, cw, cout), have_stats=have_stats)
        h8, w8 = h // 8, w // 8
        for name, src, hh, ww in (('score_conv4', 'conv4_3', h8, w8), ('score_conv5', 'conv5_3', h8 // 2, w8 // 2)):
            Z[name] = ops.conv2d_fwd(Y[src], self.w[name], P(name, 'bias'), 1, relu=False,
                                     y=self._act('z_' + name, n, hh, ww, e.Up))[0]
            Y[name] = self._bn_fwd(name, Z[name], self._act('y_' + name, n, hh, ww, e.Up))
        Z['upscore_conv5'] = ops.upsample_raw_fwd(Y['score_conv5'], 2, self._act('z_up5', n, h8, w8, e.Up))
        Y['upscore_conv5'] = self._bn_fwd('upscore_conv5', Z['upscore_conv5'], self._act('y_up5', n, h8, w8, e.Up))
        st = self.bn['upscore_conv5']
        fused = ops.upsample2x_relu_add(Y['score_conv5'], residual=Y['score_conv4'], y=self._act('fused', n, h8, w8, e.Up),
                                        scale=st.scale, shift=st.shift)            # = y_up5 + y_score_conv4
        # The x8 deconv's output (0.6 GB at 16 images) is not stored: the four batch-norm passes that would read it back
        # recompute it from `fused` per element (ops.bn_forward / bn_backward, ups8_of=) -- data-parallel runs too since round 6
        # (the statistics all-reduced between the passes: xv_bn_stats_ups8_ws); keep_all (inspection) keeps the stored map.
        virtual_up = (_VIRTUAL_UPSCORE and not self.keep_all and self.bn['upscore'].ws is not None
                      and _ups8_channels_ok(e.Up))
        score_raw = None
        if virtual_up:
            Z['upscore'] = None
            y_up = self._act('y_up', n, h, w, e.Up)
            self._bn_fwd('upscore', None, None, ups8_of=fused)                    # statistics (global under data parallelism), scale / shift
            score_raw = self._dense('score_raw', (n, h, w, e.C))
            if not ops.score_dense_fwd_ups8(fused, self.bn['upscore'], self.w['score'], P('score', 'bias'), e.C, y_up, score_raw):
                score_raw = None
                ops.bn_apply_ups8(fused, self.bn['upscore'], y_up)
            Y['upscore'] = y_up
        else:
            Z['upscore'] = ops.upsample_raw_fwd(fused, 8, self._act('z_up', n, h, w, e.Up))
            Y['upscore'] = self._bn_fwd('upscore', Z['upscore'], self._act('y_up', n, h, w, e.Up))
        if score_raw is None:
            score_raw = ops.score_dense_fwd(Y['upscore'], self.w['score'], P('score', 'bias'), e.C,
                                            self._dense('score_raw', (n, h, w, e.C)))
        mm, mv = self.moving['score']
        if self.keep_all:
            logits = ops.bn_dense_forward(score_raw, P('score', 'gamma'), P('score', 'beta'), mm, mv, self.bn['score'],
                                          self._dense('logits', (n, h, w, e.C)), sync=self._sync)
            dlogits = ops.softmax_ce_dense(logits, labels, self.count, e.C, self.loss, self._dense('dlogits', (n, h, w, e.C)))
        else:       # the normalised scores stay in registers: the loss kernel applies the batch norm's scale / shift
            ops.bn_dense_forward(score_raw, P('score', 'gamma'), P('score', 'beta'), mm, mv, self.bn['score'], None,
                                 sync=self._sync)
            dlogits = ops.softmax_ce_dense(score_raw, labels, self.count, e.C, self.loss,
                                           self._dense('dlogits', (n, h, w, e.C)), affine=self.bn['score'])
        # ---- backward ------------------------------------------------------------------------------------------
        dscore = ops.bn_dense_backward(dlogits, score_raw, P('score', 'gamma'), self.bn['score'], G('score', 'gamma'),
                                       G('score', 'beta'), self._dense('dscore', (n, h, w, e.C)), sync=self._sync)
        du = ops.score_dense_bwd(Y['upscore'], dscore, self.w['score'], e.C, G('score', 'kernel'), G('score', 'bias'),
                                 self._act('d_up', n, h, w, e.Up))
        dz_up = self._bn_bwd('upscore', du, Y['upscore'], Z['upscore'], du, ups8_of=fused if virtual_up else None)   # in place
        dfused = ops.upsample_raw_bwd(dz_up, 8, self._act('dfused', n, h8, w8, e.Up))
        wkey = ('wgrad_ws', n, h, w)
        if wkey not in self._a:
            need = 0
            for nm, _, _ in ENCODER[1:]:
                need = max(need, ops.conv2d_bwd_filter_workspace_bytes(inputs[nm], self.convs[nm][1], 3))
            for nm, src in (('score_conv4', 'conv4_3'), ('score_conv5', 'conv5_3')):
                need = max(need, ops.conv2d_bwd_filter_workspace_bytes(Y[src], e.Up, 1))
            need = max(need, ops.conv2d_first_bwd_filter_workspace_bytes(x))
            self._a[wkey] = torch.empty(need // 4, dtype=torch.float32, device=e.device)
        wws = self._a[wkey]
        # branch through score_conv4
        dz_s4 = self._bn_bwd('score_conv4', dfused, Y['score_conv4'], Z['score_conv4'], self._act('dz_s4', n, h8, w8, e.Up))
        ops.conv2d_bwd_filter(Y['conv4_3'], dz_s4, G('score_conv4', 'kernel'), G('score_conv4', 'bias'), 1, workspace=wws)
        # branch through upscore_conv5 and score_conv5
        dz_up5 = self._bn_bwd('upscore_conv5', dfused, Y['upscore_conv5'], Z['upscore_conv5'],
                              self._act('dz_up5', n, h8, w8, e.Up))
        dy_s5 = ops.upsample_raw_bwd(dz_up5, 2, self._act('dy_s5', n, h8 // 2, w8 // 2, e.Up))
        dz_s5 = self._bn_bwd('score_conv5', dy_s5, Y['score_conv5'], Z['score_conv5'], dy_s5)
        ops.conv2d_bwd_filter(Y['conv5_3'], dz_s5, G('score_conv5', 'kernel'), G('score_conv5', 'bias'), 1, workspace=wws)
        g = ops.conv2d_bwd_data(dz_s5, self.wd['score_conv5'], self.zero_bias, self._act('g_conv5_3', n, h8 // 2, w8 // 2, 512), 1)
        names = [nm for nm, _, _ in ENCODER]
        pool_after = {nm: pl for nm, _, pl in ENCODER}
        wstream = None
        if _WGRAD_STREAM and x.is_cuda:
            if getattr(self, '_wstream', None) is None:
                self._wstream = torch.cuda.Stream(device=x.device)
            wstream, main = self._wstream, torch.cuda.current_stream(x.device)
        g_is_pooled = False     # g is the gradient of the layer's POOLED output (routed inside the batch-norm passes)
        buckets_done = 0

        def bucket_ready(nm):
            # Called right behind the enqueue of layer nm's filter gradient, on the stream that carries it.  That stream is
            # ordered behind everything the bucket holds: the filter gradient waited for the event recorded on the main stream
            # after dz_<nm>, i.e. after this layer's and every earlier layer's dgamma / dbeta (main stream) and after the head's
            # gradients; earlier filter gradients precede it on the same stream.  So the bucket's all-reduce (side stream,
            # parallel.GradReducer) starts while the layers below are still differentiating, as in FcnTrainer.step.
            nonlocal buckets_done
            if reducer is not None and buckets_done < len(BN_BUCKETS) and nm == BN_BUCKETS[buckets_done][-1]:
                reducer.launch(self.grad, self.bucket_ranges[buckets_done])
                buckets_done += 1

        for nm in reversed(names):
            y = Z[nm]       # (shape only; the relu mask comes from z)
            if g_is_pooled:
                dz = ops.bn_pool_backward(g, Z[nm], self.view(self.param, nm, 'gamma'), self.bn[nm],
                                          self.view(self.grad, nm, 'gamma'), self.view(self.grad, nm, 'beta'),
                                          self._act('dz_' + nm, y.n, y.h, y.w, y.c), sync=self._sync)
            else:
                dz = self._bn_bwd(nm, g, Y[nm], Z[nm], self._act('dz_' + nm, y.n, y.h, y.w, y.c))
            g_is_pooled = False
            # the filter gradients on a second stream, as encoder_backward does (dz_<layer> and the layer inputs are buffers
            # of their own, nothing below reads the filter gradients, the slab workspace is this stream's from here on)
            if wstream is not None:
                ready = torch.cuda.Event()
                ready.record(main)
            if nm == 'conv1_1':
                with (torch.cuda.stream(wstream) if wstream is not None else contextlib.nullcontext()):
                    if wstream is not None:
                        wstream.wait_event(ready)
                    ops.conv2d_first_bwd_filter(x, dz, G(nm, 'kernel'), G(nm, 'bias'), workspace=wws)
                    bucket_ready(nm)
                break
            xin = inputs[nm]
            with (torch.cuda.stream(wstream) if wstream is not None else contextlib.nullcontext()):
                if wstream is not None:
                    wstream.wait_event(ready)
                ops.conv2d_bwd_filter(xin, dz, G(nm, 'kernel'), G(nm, 'bias'), 3, workspace=wws)
                bucket_ready(nm)
            dx = ops.conv2d_bwd_data(dz, self.wd[nm], self.zero_bias, self._act('dx_' + nm, xin.n, xin.h, xin.w, xin.c), 3)
            above = names[names.index(nm) - 1]
            if pool_after[above] and above == 'conv4_3':
                # second gradient path into conv4_3's output, through score_conv4 (AddN): the routed map is needed
                routed = ops.maxpool2x2_bwd(Y[above], dx, self._act('r_' + above, Y[above].n, Y[above].h, Y[above].w, Y[above].c))
                g = ops.conv2d_bwd_data(dz_s4, self.wd['score_conv4'], self.zero_bias,
                                        self._act('g_conv4_3', routed.n, routed.h, routed.w, routed.c), 1, addend=routed)
            elif pool_after[above]:
                g, g_is_pooled = dx, True   # MaxPoolGrad + ReluGrad happen inside the batch-norm gradient (ops.bn_pool_backward)
            else:
                g = dx
        if wstream is not None:
            main.wait_stream(wstream)
        if reducer is not None:
            assert buckets_done == len(BN_BUCKETS)
            reducer.wait()
        self.t += 1
        FcnTrainer._apply(self, 1.0)
        self.repack()
        return self.loss


# =========================================================================================================
# Training the joint two-stream model fusion_fcn (fusion_fcn.py:11-40, FusionFCN._build_graph :50-92)
# =========================================================================================================
class FusionFcnTrainer(object):
    """One training step of the joint model: per modality the relu / pool-fused VGG16 trunk of FcnTrainer (no batch
    norm), `fused_score_conv4/5` over the channel concat of the trunks' conv4_3 / conv5_3 maps, the constant x2 deconv
    + add, and decoder() with its default batch norm in training mode -- the un-commuted head of FcnBnTrainer
    (`fused/upscore` normalised between the x8 deconv and its relu, `fused/score` normalised after the 1x1 conv).

    The concat only exists in the forward pass: a 1x1 conv over concatenated channels is the sum of one 1x1 conv per
    modality, so the filter gradient is taken per 512-row block of the kernel and the data gradient per modality with
    that block's weights, straight into the trunk walk (encoder_backward)."""

    HEAD = ('score', 'upscore')            # decoder layers with batch norm (scope `fused/`)

    def __init__(self, engine, trainer='rmsprop', learning_rate=1e-4):
        if trainer not in ('adam', 'rmsprop', 'adagrad'):
            raise KeyError(trainer)
        self.e, self.kind, self.lr = engine, trainer, float(learning_rate)
        e, dev = engine, engine.device
        if not _ups8_channels_ok(e.Up):
            raise ValueError('fusion_fcn training: num_units = %d is padded to %d lanes; the decoder\'s batch-norm kernels take '
                             '64, 128 or 256 (num_units <= 256)' % (e.U, e.Up))
        self.mods = list(e.prefixes)
        nm = len(self.mods)
        entries = [(('score', 'kernel'), (e.Up, e.C)), (('score', 'bias'), (e.C,)), (('score', 'gamma'), (e.C,)),
                   (('score', 'beta'), (e.C,)), (('upscore', 'gamma'), (e.Up,)), (('upscore', 'beta'), (e.Up,))]
        for name in ('fused_score_conv5', 'fused_score_conv4'):
            entries += [((name, 'kernel'), (1, 1, 512 * nm, e.Up)), ((name, 'bias'), (e.Up,))]
        self.bucket_ranges, total = [], 0
        self.offsets = {}

        def place(items):
            nonlocal total
            b0 = total
            for key, shape in items:
                n = int(np.prod(shape))
                self.offsets[key] = (total, n, shape)
                total += (n + 63) // 64 * 64
            self.bucket_ranges.append((b0, total))

        place(entries)
        for m in self.mods:                # one bucket per trunk, in backward order
            cins = [e.num_channels[m]] + [c for _, c, _ in ENCODER[:-1]]
            items = []
            for (name, cout, _), cin in reversed(list(zip(ENCODER, cins))):
                items += [(((m, name), 'kernel'), (3, 3, cin, cout)), (((m, name), 'bias'), (cout,))]
            place(items)
        self.total = total
        self.param = torch.zeros(total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.bn_channels = {'score': e.C, 'upscore': e.Up}
        self.moving = {k: (torch.zeros(c, device=dev), torch.ones(c, device=dev)) for k, c in self.bn_channels.items()}
        self.bn = {k: ops.BnState(c, dev) for k, c in self.bn_channels.items()}
        self.state, self.t = {}, 0
        self.zero_bias = torch.zeros(512, dtype=torch.float32, device=dev)
        self.count = torch.zeros(1, dtype=torch.int64, device=dev)
        self.loss = torch.zeros(1, dtype=torch.float64, device=dev)
        self.wd, self._a, self._sync = {}, {}, False

    def view(self, buf, name, kind):
        off, n, shape = self.offsets[(name, kind)]
        return buf[off:off + n].view(*shape)

    # ---- variables <-> master buffer ---------------------------------------------------------------------
    def _names(self):
        """(key in the master buffer, kind, variable name in the reference schema, real leading/trailing slice)."""
        e = self.e
        for m in self.mods:
            for name, _, _ in ENCODER:
                for kind in ('kernel', 'bias'):
                    yield (m, name), kind, '%s_%s/%s' % (e.prefixes[m], name, kind)
        for name in ('fused_score_conv4', 'fused_score_conv5'):
            for kind in ('kernel', 'bias'):
                yield name, kind, '%s/%s' % (name, kind)
        for kind in ('kernel', 'bias', 'gamma', 'beta'):
            yield 'score', kind, 'fused/score/' + kind
        for kind in ('gamma', 'beta'):
            yield 'upscore', kind, 'fused/upscore/' + kind

    def _real(self, key, kind, t):
        """View of the channels that exist in the reference graph (U of the Up padded lanes)."""
        e = self.e
        if key == 'score' and kind == 'kernel':
            return t[:e.U]
        if key in ('fused_score_conv4', 'fused_score_conv5') or (key == 'upscore'):
            return t[..., :e.U]
        return t

    def load_from_variables(self, variables):
        e = self.e
        self.param.zero_()
        self.view(self.param, 'upscore', 'gamma').fill_(1.0)     # padding lanes: gamma 1, beta 0, maps stay zero
        for key, kind, vname in self._names():
            src = torch.from_numpy(np.asarray(variables[vname], np.float32))
            dst = self._real(key, kind, self.view(self.param, key, kind))
            dst.copy_(src.reshape(dst.shape))
        for key, layer in (('score', 'fused/score'), ('upscore', 'fused/upscore')):
            mm, mv = self.moving[key]
            mm.zero_()
            mv.fill_(1.0)
            real = e.C if key == 'score' else e.U
            mm[:real].copy_(torch.from_numpy(np.asarray(variables[layer + '/moving_mean'], np.float32)))
            mv[:real].copy_(torch.from_numpy(np.asarray(variables[layer + '/moving_variance'], np.float32)))
        self.repack()

    def _export(self, buf):
        out = {}
        for key, kind, vname in self._names():
            t = self._real(key, kind, self.view(buf, key, kind)).cpu().numpy()
            if key == 'score' and kind == 'kernel':
                t = t.reshape(1, 1, self.e.U, self.e.C)
            out[vname] = np.ascontiguousarray(t)
        return out

    def to_variables(self, variables):
        variables.update(self._export(self.param))
        for key, layer in (('score', 'fused/score'), ('upscore', 'fused/upscore')):
            real = self.e.C if key == 'score' else self.e.U
            variables[layer + '/moving_mean'] = self.moving[key][0][:real].cpu().numpy().copy()
            variables[layer + '/moving_variance'] = self.moving[key][1][:real].cpu().numpy().copy()

    def grads_as_variables(self):
        return self._export(self.grad)

    def repack(self):
        """Master fp32 weights -> packed bf16 forward weights inside the engine's trunks, packed data-gradient weights
        here (per modality for the 512-row blocks of the fused 1x1 kernels)."""
        e, dev = self.e, self.e.device

        def packed(store, key, kv, dgrad):
            if key not in store:
                k, _, cin, cout = kv.shape
                store[key] = torch.empty(ops.packed_weight_elems(k, cin, cout), dtype=torch.bfloat16, device=dev)
            (ops.pack_conv_weights_dgrad if dgrad else ops.pack_conv_weights_into)(kv, store[key])

        # the 2 x 12 trunk kernels: ONE launch over a descriptor table (as FcnTrainer.repack; the master views and the packed
        # buffers never move -- a trunk whose load() installed buffers of its own gets a new table)
        entries = []
        stale = getattr(self, '_pack_table', None) is None
        for m in self.mods:
            trunk = e.trunks[m]
            for name, _, _ in ENCODER:
                kv = self.view(self.param, (m, name), 'kernel')
                trunk.b[name] = self.view(self.param, (m, name), 'bias')
                if name == 'conv1_1':
                    trunk.w[name] = kv
                    continue
                if not getattr(trunk, '_trainer_owned', False) or name not in trunk.w:
                    trunk.w[name] = torch.empty(ops.packed_weight_elems(3, kv.shape[2], kv.shape[3]), dtype=torch.bfloat16,
                                                device=dev)
                    stale = True
                if (m, name) not in self.wd:
                    self.wd[(m, name)] = torch.empty_like(trunk.w[name])
                    stale = True
                entries.append((kv, trunk.w[name], self.wd[(m, name)]))
            trunk._trainer_owned = True
        if stale:
            self._pack_table = ops.PackTable(entries, dev)
        self._pack_table.run()
        for name in ('fused_score_conv4', 'fused_score_conv5'):
            kv = self.view(self.param, name, 'kernel')
            e.b[name] = self.view(self.param, name, 'bias')
            packed(e.w, name, kv, False)
            for i, m in enumerate(self.mods):
                packed(self.wd, (m, name), kv[:, :, 512 * i:512 * (i + 1), :].contiguous(), True)

    # ---- scratch -----------------------------------------------------------------------------------------------
    def _act(self, tag, n, h, w, c):
        key = (tag, n, h, w, c)
        a = self._a.get(key)
        if a is None:
            a = self._a[key] = ops.Act(n, h, w, c, self.e.device)
        return a

    def _gact(self, like, tag):
        return self._act(tag, like.n, like.h, like.w, like.c)

    def _dense(self, tag, shape):
        key = (tag,) + tuple(shape)
        t = self._a.get(key)
        if t is None:
            t = self._a[key] = torch.empty(shape, dtype=torch.float32, device=self.e.device)
        return t

    # ---- one training step ----------------------------------------------------------------------------------
    def step(self, inputs, labels, reducer=None):
        """inputs: {modality: float32 [N,H,W,c]}, labels: int32 [N,H,W] (device tensors) -> loss (device float64)."""
        e = self.e
        P = lambda name, kind: self.view(self.param, name, kind)   # noqa: E731
        G = lambda name, kind: self.view(self.grad, name, kind)    # noqa: E731
        x0 = inputs[self.mods[0]]
        n, h, w, _ = x0.shape
        h8, w8 = h // 8, w // 8
        self.grad.zero_()
        self.loss.zero_()
        self.count.zero_()
        ops.count_valid_labels(labels, e.C, self.count)
        self._sync = reducer is not None
        if reducer is not None:
            reducer.allreduce_now(self.count)
        # ---- forward ---------------------------------------------------------------------------------------------
        L = {m: e.trunks[m].forward(inputs[m], keep_all=True, routed=True) for m in self.mods}

        def concat(name):
            cur = L[self.mods[0]][name]
            for i, m in enumerate(self.mods[1:]):
                nxt = L[m][name]
                cur = ops.concat_channels(cur, nxt, self._act('cat_%s_%d' % (name, i), cur.n, cur.h, cur.w, cur.c + nxt.c))
            return cur

        s4 = ops.conv2d_fwd(concat('conv4_3'), e.w['fused_score_conv4'], e.b['fused_score_conv4'], 1, relu=True,
                            y=self._act('s4', n, h8, w8, e.Up))[0]
        s5 = ops.conv2d_fwd(concat('conv5_3'), e.w['fused_score_conv5'], e.b['fused_score_conv5'], 1, relu=True,
                            y=self._act('s5', n, h8 // 2, w8 // 2, e.Up))[0]
        feat = ops.upsample2x_relu_add(s5, residual=s4, y=self._act('features', n, h8, w8, e.Up))
        # (the x8 deconv's output is not stored: see FcnBnTrainer.step)
        virtual_up = _VIRTUAL_UPSCORE and self.bn['upscore'].ws is not None and _ups8_channels_ok(e.Up)
        mm, mv = self.moving['upscore']
        z_up = None if virtual_up else ops.upsample_raw_fwd(feat, 8, self._act('z_up', n, h, w, e.Up))
        score_raw = None
        if virtual_up:
            y_up = self._act('y_up', n, h, w, e.Up)
            ops.bn_forward(None, P('upscore', 'gamma'), P('upscore', 'beta'), mm, mv, self.bn['upscore'], None, relu=True,
                           sync=self._sync, ups8_of=feat)                           # statistics, scale / shift
            score_raw = self._dense('score_raw', (n, h, w, e.C))
            if not ops.score_dense_fwd_ups8(feat, self.bn['upscore'], P('score', 'kernel'), P('score', 'bias'), e.C, y_up, score_raw):
                score_raw = None
                ops.bn_apply_ups8(feat, self.bn['upscore'], y_up)
        else:
            y_up = ops.bn_forward(z_up, P('upscore', 'gamma'), P('upscore', 'beta'), mm, mv, self.bn['upscore'],
                                  self._act('y_up', n, h, w, e.Up), relu=True, sync=self._sync)
        if score_raw is None:
            score_raw = ops.score_dense_fwd(y_up, P('score', 'kernel'), P('score', 'bias'), e.C,
                                            self._dense('score_raw', (n, h, w, e.C)))
        mm, mv = self.moving['score']
        logits = ops.bn_dense_forward(score_raw, P('score', 'gamma'), P('score', 'beta'), mm, mv, self.bn['score'],
                                      self._dense('logits', (n, h, w, e.C)), sync=self._sync)
        dlogits = ops.softmax_ce_dense(logits, labels, self.count, e.C, self.loss, self._dense('dlogits', (n, h, w, e.C)))
        # ---- backward: head ------------------------------------------------------------------------------------
        dscore = ops.bn_dense_backward(dlogits, score_raw, P('score', 'gamma'), self.bn['score'], G('score', 'gamma'),
                                       G('score', 'beta'), self._dense('dscore', (n, h, w, e.C)), sync=self._sync)
        du = ops.score_dense_bwd(y_up, dscore, P('score', 'kernel'), e.C, G('score', 'kernel'), G('score', 'bias'),
                                 self._act('d_up', n, h, w, e.Up))
        dz_up = ops.bn_backward(du, y_up, z_up, P('upscore', 'gamma'), self.bn['upscore'], G('upscore', 'gamma'),
                                G('upscore', 'beta'), du, sync=self._sync, ups8_of=feat if virtual_up else None)
        dfeat = ops.upsample_raw_bwd(dz_up, 8, self._act('dfeat', n, h8, w8, e.Up))
        ds4 = ops.relu_bwd(dfeat, s4, self._act('ds4', n, h8, w8, e.Up))
        ds5 = ops.upsample2x_bwd(dfeat, s5, self._act('ds5', n, h8 // 2, w8 // 2, e.Up))
        wkey = ('wgrad_ws', n, h, w)
        if wkey not in self._a:
            need = 0
            for m in self.mods:
                prev = None
                for nm, cout, pool in ENCODER:
                    if prev is not None:
                        need = max(need, ops.conv2d_bwd_filter_workspace_bytes(L[m][prev], cout, 3))
                    prev = pool if pool else nm
                for src in ('conv4_3', 'conv5_3'):
                    need = max(need, ops.conv2d_bwd_filter_workspace_bytes(L[m][src], e.Up, 1))
                need = max(need, ops.conv2d_first_bwd_filter_workspace_bytes(inputs[m]))
            self._a[wkey] = torch.empty(max(need // 4, 1), dtype=torch.float32, device=e.device)
        wws = self._a[wkey]
        # ---- backward: fused 1x1 convs block by block, then each trunk ----------------------------------------------
        for i, m in enumerate(self.mods):
            rows = slice(512 * i, 512 * (i + 1))
            first = i == 0                                      # the bias gradient is the same sum for every block
            ops.conv2d_bwd_filter(L[m]['conv4_3'], ds4, G('fused_score_conv4', 'kernel')[:, :, rows, :],
                                  G('fused_score_conv4', 'bias') if first else None, 1, workspace=wws)
            ops.conv2d_bwd_filter(L[m]['conv5_3'], ds5, G('fused_score_conv5', 'kernel')[:, :, rows, :],
                                  G('fused_score_conv5', 'bias') if first else None, 1, workspace=wws)
        if reducer is not None:
            reducer.launch(self.grad, self.bucket_ranges[0])
        for i, m in enumerate(self.mods):
            Lm = L[m]
            g = ops.conv2d_bwd_data(ds5, self.wd[(m, 'fused_score_conv5')], self.zero_bias,
                                    self._act('g5_' + m, n, h8 // 2, w8 // 2, 512), 1, relu_ref=Lm['conv5_3'])
            wd = {name: self.wd[(m, name)] for name, _, _ in ENCODER[1:]}
            Gm = lambda name, kind, m=m: self.view(self.grad, (m, name), kind)              # noqa: E731
            gact = lambda like, tag, m=m: self._gact(like, '%s_%s' % (tag, m))              # noqa: E731
            if _WGRAD_STREAM and inputs[m].is_cuda and getattr(self, '_wstream', None) is None:
                self._wstream = torch.cuda.Stream(device=inputs[m].device)
            encoder_backward(inputs[m], Lm, g, ds4, self.wd[(m, 'fused_score_conv4')], wd, self.zero_bias, Gm, gact, wws,
                             wstream=self._wstream if _WGRAD_STREAM and inputs[m].is_cuda else None)
            if reducer is not None:
                reducer.launch(self.grad, self.bucket_ranges[1 + i])
        if reducer is not None:
            reducer.wait()
        self.t += 1
        FcnTrainer._apply(self, 1.0)
        self.repack()
        return self.loss
