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
import numpy as np
import torch

from . import ops
from .fcn import ENCODER, variable_shapes

# backward order of the trainable layers
LAYER_ORDER = ['score', 'score_conv5', 'score_conv4'] + [name for name, _, _ in reversed(ENCODER)]
BUCKETS = [['score', 'score_conv5', 'score_conv4', 'conv5_3', 'conv5_2', 'conv5_1'],
           ['conv4_3', 'conv4_2', 'conv4_1'],
           ['conv3_3', 'conv3_2', 'conv3_1', 'conv2_2', 'conv2_1', 'conv1_2', 'conv1_1']]


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
        """Master fp32 weights -> the engine's bf16 packed forward weights + packed dgrad weights."""
        e = self.e
        for name in LAYER_ORDER:
            kv, bv = self.view(self.param, name, 'kernel'), self.view(self.param, name, 'bias')
            e.b[name] = bv
            if name in ('score', 'conv1_1'):
                e.w[name] = kv
                continue
            if name not in self.wd:
                k, _, cin, cout = kv.shape
                nel = ops.packed_weight_elems(k, cin, cout)
                e.w[name] = torch.empty(nel, dtype=torch.bfloat16, device=e.device)
                self.wd[name] = torch.empty(nel, dtype=torch.bfloat16, device=e.device)
            ops.pack_conv_weights_into(kv, e.w[name])
            ops.pack_conv_weights_dgrad(kv, self.wd[name])

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
        L = e.encoder(x, keep_all=True)
        n, h, w, _ = x.shape
        self.grad.zero_()
        self.loss.zero_()
        self.count.zero_()
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
        # walk the encoder backwards
        names = [nm for nm, _, _ in ENCODER]
        pool_after = {nm: pl for nm, _, pl in ENCODER}
        inputs = {}
        prev = None
        for nm in names:
            inputs[nm] = prev
            prev = pool_after[nm] if pool_after[nm] else nm
        done_buckets = 0
        for nm in reversed(names):
            xin = inputs[nm]
            if nm == 'conv1_1':
                ops.conv2d_first_bwd_filter(x, g, G(nm, 'kernel'), G(nm, 'bias'))
            else:
                ops.conv2d_bwd_filter(L[xin], g, G(nm, 'kernel'), G(nm, 'bias'), 3, workspace=wws)
            if reducer is not None and nm == BUCKETS[done_buckets][-1]:
                reducer.launch(self.grad, self.bucket_ranges[done_buckets])
                done_buckets += 1
            if nm == 'conv1_1':
                break
            if xin.startswith('pool'):
                # gradient w.r.t. the pooled map, then MaxPoolGrad + ReluGrad onto the conv above
                dpool = ops.conv2d_bwd_data(g, self.wd[nm], self.zero_bias, self._gact(L[xin], 'g_' + xin), 3)
                above = names[names.index(nm) - 1]
                routed = ops.maxpool2x2_bwd(L[above], dpool, self._gact(L[above], 'r_' + above))
                if above == 'conv4_3':
                    # second gradient path into conv4_3: through score_conv4 (AddN), then its relu
                    g = ops.conv2d_bwd_data(ds4, self.wd['score_conv4'], self.zero_bias,
                                            self._gact(L[above], 'g_' + above), 1, relu_ref=L[above], addend=routed)
                else:
                    g = routed
            else:
                g = ops.conv2d_bwd_data(g, self.wd[nm], self.zero_bias, self._gact(L[xin], 'g_' + xin), 3,
                                        relu_ref=L[xin])
        scale = 1.0
        if reducer is not None:
            reducer.wait()
            scale = 1.0                                 # gradients are sums over the global batch / global denominator
        self.t += 1
        self._apply(scale)
        self.repack()
        return self.loss

    def _apply(self, scale):
        if self.kind == 'adam':
            if not self.state:
                self.state = {'m': torch.zeros_like(self.param), 'v': torch.zeros_like(self.param)}
            b1, b2 = 0.9, 0.999
            lr_t = self.lr * np.sqrt(1 - b2 ** self.t) / (1 - b1 ** self.t)
            ops.adam_step(self.param, self.grad, self.state['m'], self.state['v'], float(lr_t), b1, b2, 1e-8, scale)
        elif self.kind == 'rmsprop':
            if not self.state:
                self.state = {'ms': torch.ones_like(self.param)}
            ops.rmsprop_step(self.param, self.grad, self.state['ms'], self.lr, 0.9, 1e-10, scale)
        else:
            if not self.state:
                self.state = {'acc': torch.full_like(self.param, 0.1)}
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
