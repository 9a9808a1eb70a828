"""One training step of the AdapNet expert on MI355X: adapnet(..., is_training=True) (adapnet.py:103-173), the loss of
Adapnet._build_graph (adapnet.py:196-203) and the [TF1] optimizers of base_model.py:153-162.

Every conv / deconv of the graph is followed by tf.layers.batch_normalization(training=True), so nothing folds: a
unit is conv -> z (stored) -> batch statistics -> y = [relu](BN(z)) (stored), and its backward is the batch-norm
gradient (which carries the relu mask), the MFMA filter gradient and the MFMA data gradient.  The graph is a DAG
(block inputs feed the first stage and the shortcut, block 7 feeds block 8 and the `shortcut` conv), so the step
records a tape of backward closures and accumulates gradients per tensor name.  The strided / atrous convs run
through the same gathers as the inference engine (adapnet.AdapnetEngine); their weights are DERIVED tensors (the 7x7
kernel scattered into 3x3 x 9 groups, the atrous pair stacked block-wise), rebuilt from the master weights every step,
and their filter gradients are mapped back through the same index maps.

The two deconvs are TRAINED, as in the reference: adapnet.py:155-163 calls custom_layers.deconv2d (:71-121) without
trainable=False, so `first_deconvolution_upconv/kernel` [4,4,U,2048] and `second_deconvolution_upconv/kernel` [16,16,C,U]
start as the bilinear constant and are dense after the first optimizer step.  Each runs as ONE 3x3 MFMA conv onto
stride^2 phase channels at its input resolution (the derived kernel custom_layers.dense_deconv_as_conv3x3 arranges,
rebuilt from the master kernel every step like the other derived kernels) + a depth-to-space shuffle; its filter
gradient is the 3x3 conv's filter gradient against the space-to-depth shuffle of the upstream gradient, gathered back
through the inverse index map (every kernel element sits at exactly one derived position), its data gradient the 3x3
conv's data gradient.  `first_deconvolution_conv` is therefore trained on all of its output channels.  The loss is the
reference's: the mean cross-entropy over the labelled pixels divided once more by their number (adapnet.py:202-203).
"""
import os

import numpy as np
import torch

from . import ops
from .adapnet import _conv_scopes, conv7s2_as_3x3
from .custom_layers import dense_deconv_as_conv3x3
from .trainer import FcnTrainer

_IMPLICIT_PAIRS = os.environ.get('XV_IMPLICIT_PAIRS', '1') != '0'   # 0: the atrous pairs through the im2col operand (A/B)
_FUSE_ACCUM = os.environ.get('XV_ADAPNET_FUSE_ACCUM', '1') != '0'   # 0: every second gradient of a tensor through an add pass (A/B)


def conv7s2_index_maps(cin, cout):
    """Index maps between the 7x7 stride-2 kernel [7,7,cin,cout] and the derived [3,3,9*cin,cout] kernel of
    adapnet.conv7s2_as_3x3 (both flattened): src[j] = flat index of the 7x7 element that derived element j copies, -1
    where the derived kernel is structurally zero; inv[i] = the derived position of 7x7 element i (each is used exactly
    once).  derived = w7.ravel()[src] (masked), and the filter gradient maps back as dw7.ravel() = dw3.ravel()[inv]."""
    n = 49 * cin * cout
    assert n < 2 ** 24                                    # the probe values below are exact in float32
    probe = np.arange(1, n + 1, dtype=np.float32).reshape(7, 7, cin, cout)
    src = conv7s2_as_3x3(probe).astype(np.int64).ravel() - 1
    inv = np.empty(n, np.int64)
    pos = np.nonzero(src >= 0)[0]
    inv[src[pos]] = pos
    return src, inv


def dense_deconv_index_maps(shape, stride):
    """The same pair of maps between a (channel-padded) transposed-conv kernel [k,k,F,Cin] and the derived 3x3 kernel
    [3,3,Cin,stride^2*F] of custom_layers.dense_deconv_as_conv3x3: derived = w.ravel()[src] (masked where src < 0), and
    dw.ravel() = dderived.ravel()[inv]."""
    n = int(np.prod(shape))
    assert n < 2 ** 24
    probe = np.arange(1, n + 1, dtype=np.float32).reshape(shape)
    src = dense_deconv_as_conv3x3(probe, stride).astype(np.int64).ravel() - 1
    inv = np.empty(n, np.int64)
    pos = np.nonzero(src >= 0)[0]
    assert len(pos) == n                                  # k = 2 * stride: every kernel element is used exactly once
    inv[src[pos]] = pos
    return src, inv


DECONVS = (('first_deconvolution_upconv', 2), ('second_deconvolution_upconv', 8))


class AdapnetTrainer(object):
    def __init__(self, engine, trainer='adam', learning_rate=1e-4):
        if trainer not in ('adam', 'rmsprop', 'adagrad'):
            raise KeyError(trainer)
        self.e, self.kind, self.lr = engine, trainer, float(learning_rate)
        e, dev = engine, engine.device
        # ---- units: name -> (k, cin, cout as trained, has_bias) ----------------------------------------------------
        self.units = {}
        self.blocks = e.blocks
        for scope, k, cin, cout, has_bias in _conv_scopes(e.cin, e.U, self.blocks):
            if scope == 'shortcut':
                cout = e.Up                                   # U padded to 64 lanes (zero kernels, gamma 1, beta 0)
            self.units[scope] = (k, cin, cout, has_bias)
        self.width = self.units['first_deconvolution_conv'][2]         # 2048 in the reference graph
        self.Cp = (e.C + 7) // 8 * 8
        # the two trainable transposed-conv kernels [k,k,filters,in], channel-padded like the maps they touch
        self.deconv_shape = {'first_deconvolution_upconv': (4, 4, e.Up, self.width),
                             'second_deconvolution_upconv': (16, 16, self.Cp, e.Up)}
        self.deconv_real = {'first_deconvolution_upconv': (e.U, self.width), 'second_deconvolution_upconv': (e.C, e.U)}
        self.bn_channels = {scope: u[2] for scope, u in self.units.items()}
        self.bn_channels.update(first_deconvolution_upconv=e.Up, second_deconvolution_upconv=e.C)
        self.offsets, total = {}, 0
        for scope, (k, cin, cout, has_bias) in self.units.items():
            entries = [('kernel', (k, k, cin, cout))] + ([('bias', (cout,))] if has_bias else [])
            for kind, shape in entries + [('gamma', (cout,)), ('beta', (cout,))]:
                n = int(np.prod(shape))
                self.offsets[(scope, kind)] = (total, n, shape)
                total += (n + 63) // 64 * 64
        for scope in ('first_deconvolution_upconv', 'second_deconvolution_upconv'):
            n = int(np.prod(self.deconv_shape[scope]))
            self.offsets[(scope, 'kernel')] = (total, n, self.deconv_shape[scope])
            total += (n + 63) // 64 * 64
            for kind in ('gamma', 'beta'):
                n = self.bn_channels[scope]
                self.offsets[(scope, kind)] = (total, n, (n,))
                total += (n + 63) // 64 * 64
        self.total = total
        self.param = torch.zeros(total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.moving = {s: (torch.zeros(c, device=dev), torch.ones(c, device=dev)) for s, c in self.bn_channels.items()}
        self.bn = {s: ops.BnState(c, dev) for s, c in self.bn_channels.items()}
        self.state, self.t = {}, 0
        self.zeros = torch.zeros(9216, dtype=torch.float32, device=dev)
        self.count = torch.zeros(1, dtype=torch.int64, device=dev)
        self.loss = torch.zeros(1, dtype=torch.float64, device=dev)
        self.grad_scale = 1.0
        self.w, self.wd, self._a, self._scratch = {}, {}, {}, {}
        self.wd_pair = {}      # block_b name -> packed image of the pair's implicit data gradient
        self.dmap = {}
        for scope, stride in DECONVS:
            src, inv = dense_deconv_index_maps(self.deconv_shape[scope], stride)
            idx = torch.from_numpy(src).to(dev)
            k, _, f, cin = self.deconv_shape[scope]
            self.dmap[scope] = (idx.clamp(min=0), (idx >= 0), torch.from_numpy(inv).to(dev), (3, 3, cin, stride * stride * f))
        src, inv = conv7s2_index_maps(64, 64)
        idx = torch.from_numpy(src).to(dev)
        self.map7 = (idx.clamp(min=0), (idx >= 0))
        self.inv7 = torch.from_numpy(inv).to(dev)
        self._sync = False

    # ---- parameters ------------------------------------------------------------------------------------------------
    def view(self, buf, name, kind):
        off, n, shape = self.offsets[(name, kind)]
        return buf[off:off + n].view(*shape)

    def _real(self, scope):
        e = self.e
        if scope in ('shortcut', 'first_deconvolution_upconv'):
            return e.U
        return self.bn_channels[scope]

    def load_from_variables(self, variables):
        p = self.e.prefix
        self.param.zero_()
        for (scope, kind), _ in self.offsets.items():
            real = self._real(scope)
            dst = self.view(self.param, scope, kind)
            if kind == 'gamma':
                dst.fill_(1.0)
            src = torch.from_numpy(np.asarray(variables['%s/%s/%s' % (p, scope, kind)], np.float32))
            if kind == 'kernel' and scope in self.deconv_shape:
                f, cin = self.deconv_real[scope]               # [k,k,filters,in]: both channel axes are padded
                dst[:, :, :f, :cin].copy_(src)
                continue
            dst[..., :real].copy_(src[..., :real])
        for scope, (mm, mv) in self.moving.items():
            real = self._real(scope)
            mm.zero_()
            mv.fill_(1.0)
            mm[:real].copy_(torch.from_numpy(np.asarray(variables['%s/%s/moving_mean' % (p, scope)], np.float32))[:real])
            mv[:real].copy_(torch.from_numpy(np.asarray(variables['%s/%s/moving_variance' % (p, scope)], np.float32))[:real])
        self.repack()

    def _export(self, buf, variables, scale=1.0):
        p = self.e.prefix
        for (scope, kind), _ in self.offsets.items():
            real = self._real(scope)
            name = '%s/%s/%s' % (p, scope, kind)
            if kind == 'kernel' and scope in self.deconv_shape:
                f, cin = self.deconv_real[scope]
                t = (self.view(buf, scope, kind)[:, :, :f, :cin] * scale).cpu().numpy()
            else:
                t = (self.view(buf, scope, kind)[..., :real] * scale).cpu().numpy()
            variables[name] = np.ascontiguousarray(t)
        return variables

    def to_variables(self, variables):
        p = self.e.prefix
        self._export(self.param, variables)
        for scope, (mm, mv) in self.moving.items():
            real = self._real(scope)
            for v, src in (('moving_mean', mm), ('moving_variance', mv)):
                name = '%s/%s/%s' % (p, scope, v)
                t = src[:real].cpu().numpy().copy()
                if name in variables and np.shape(variables[name])[-1] != real:
                    full = np.array(variables[name], np.float32, copy=True)
                    full[:real] = t
                    t = full
                variables[name] = t

    def grads_as_variables(self):
        """Gradient of the last step in the reference schema, including the 1/count of the loss (tests)."""
        return self._export(self.grad, {}, scale=self.grad_scale)

    def _kernel_for(self, scope):
        """The kernel the MFMA convs consume for `scope` (derived for the 7x7 stride-2 conv)."""
        kv = self.view(self.param, scope, 'kernel')
        if scope == 'block_0_2':
            src, mask = self.map7
            return (kv.reshape(-1)[src] * mask).view(3, 3, 9 * 64, 64)
        if scope in self.dmap:
            src, mask, _, shape = self.dmap[scope]
            return (kv.reshape(-1)[src] * mask).view(*shape)
        return kv

    def _pair_kernel(self, name):
        k1, k2 = self.view(self.param, name + '/stage_2_1', 'kernel'), self.view(self.param, name + '/stage_2_2', 'kernel')
        c, half = k1.shape[2], k1.shape[3]
        w = self._scratch.get(('pairw', name))
        if w is None:
            w = self._scratch[('pairw', name)] = torch.zeros(1, 1, 18 * c, 2 * half, dtype=torch.float32, device=k1.device)
        w[0, 0, :9 * c, :half] = k1.reshape(9 * c, half)
        w[0, 0, 9 * c:, half:] = k2.reshape(9 * c, half)
        return w

    def _pack(self, key, kernel):
        if key not in self.w:
            k, _, cin, cout = kernel.shape
            nel = ops.packed_weight_elems(k, cin, cout)
            self.w[key] = torch.empty(nel, dtype=torch.bfloat16, device=kernel.device)
            self.wd[key] = torch.empty(nel, dtype=torch.bfloat16, device=kernel.device)
        kernel = kernel.contiguous()
        ops.pack_conv_weights_pair(kernel, self.w[key], self.wd[key])

    def repack(self):
        """Master fp32 kernels -> packed forward / data-gradient images.  The kernels that are plain views of the flat
        parameter tensor, and the block-diagonal pair kernels (scratch tensors that never move), go through ONE launch over
        a descriptor table (ops.PackTable, as FcnTrainer.repack: ~50 launches of 8 us less per step); the three derived
        kernels (the 7x7 stride-2 conv, the two deconvs) are gathered first and packed one by one."""
        table = getattr(self, '_pack_table', None)
        entries = [] if table is None else None
        for scope in list(self.units) + [d[0] for d in DECONVS]:
            if scope == 'block_0_1' or '/stage_2_' in scope:
                continue
            if scope == 'block_0_2' or scope in self.dmap:
                kernel = self._kernel_for(scope)
                self._pack(scope, kernel)
                if scope == 'second_deconvolution_upconv':
                    # the class scores are computed in float32 (ops.deconv8_scores_f32): keep the derived 3x3 kernel as it is
                    self.w32_scores = kernel.contiguous()
            elif entries is not None:
                entries.append((scope, self.view(self.param, scope, 'kernel')))
        pair_entries = []
        for name, kind, args in self.blocks:
            if kind == 'b':
                w = self._pair_kernel(name)             # (two slice copies into a scratch tensor of its own)
                if entries is not None:
                    entries.append((name + '/stage_2', w))
                f1, f2 = args[0], args[1]
                if _IMPLICIT_PAIRS and f1 % 256 == 0 and f2 % 256 == 0:
                    # the [1,1,18 f2/2,f1] kernel of the pair's implicit data gradient (ops.dilated_pair_dgrad_kernel), packed
                    # in the forward format: a scratch tensor of its own too
                    wdk = ops.dilated_pair_dgrad_kernel(self.view(self.param, name + '/stage_2_1', 'kernel'),
                                                        self.view(self.param, name + '/stage_2_2', 'kernel'),
                                                        out=self._scratch.get(('pairwd', name)))
                    self._scratch[('pairwd', name)] = wdk
                    if entries is not None:
                        self.wd_pair[name] = torch.empty(ops.packed_weight_elems(1, wdk.shape[2], wdk.shape[3]), dtype=torch.bfloat16,
                                                         device=wdk.device)
                        pair_entries.append((wdk, self.wd_pair[name], None))
        if entries is not None:
            triples = []
            for key, kernel in entries:
                k, _, cin, cout = kernel.shape
                if not kernel.is_contiguous():
                    raise ValueError('kernel view of %s is not contiguous' % key)
                nel = ops.packed_weight_elems(k, cin, cout)
                self.w[key] = torch.empty(nel, dtype=torch.bfloat16, device=kernel.device)
                self.wd[key] = torch.empty(nel, dtype=torch.bfloat16, device=kernel.device)
                triples.append((kernel, self.w[key], self.wd[key]))
            table = self._pack_table = ops.PackTable(triples + pair_entries, self.param.device)
        table.run()

    # ---- scratch ---------------------------------------------------------------------------------------------------
    def _act(self, tag, n, h, w, c):
        key = (tag, n, h, w, c)
        a = self._a.get(key)
        if a is None:
            a = self._a[key] = ops.Act(n, h, w, c, self.e.device)
        return a

    def _like(self, tag, a):
        return self._act(tag, a.n, a.h, a.w, a.c)

    def _dense(self, tag, shape):
        key = (tag,) + tuple(shape)
        t = self._a.get(key)
        if t is None:
            t = self._a[key] = torch.empty(shape, dtype=torch.float32, device=self.e.device)
        return t

    # ---- gradient bookkeeping ------------------------------------------------------------------------------------
    def _accum(self, name, g):
        """Add g to the gradient of tensor `name` (first contribution: keep a reference; later ones: into a buffer this
        bookkeeping owns, since the first reference may be shared with another tensor's gradient)."""
        cur = self._grads.get(name)
        if cur is None:
            self._grads[name] = (g, False)
            return
        buf, owned = cur
        if not owned:
            own = self._like('gsum_' + name, buf)
            ops.add(buf, g, own)
            self._grads[name] = (own, True)
        else:
            ops.add(buf, g, buf)

    def _bn_unit(self, scope, z, relu, name):
        mm, mv = self.moving[scope]
        P = lambda kind: self.view(self.param, scope, kind)        # noqa: E731
        G = lambda kind: self.view(self.grad, scope, kind)         # noqa: E731
        y = ops.bn_forward(z, P('gamma'), P('beta'), mm, mv, self.bn[scope], self._like('y_' + name, z), relu=relu,
                           sync=self._sync)

        def bwd_bn(dy):
            return ops.bn_backward(dy, y if relu else None, z, P('gamma'), self.bn[scope], G('gamma'), G('beta'),
                                   self._like('dz_' + name, z), sync=self._sync)
        return y, bwd_bn

    # ---- one training step ----------------------------------------------------------------------------------------
    def step(self, x, labels, reducer=None):
        """x: float32 [N,H,W,cin], labels: int32 [N,H,W] (device tensors) -> loss (device float64 scalar)."""
        e = self.e
        n, h, w, _ = x.shape
        if h % 16 or w % 16:
            raise ValueError('H and W must be multiples of 16')
        ops.zero_(self.grad)
        ops.zero_(self.loss)
        ops.zero_(self.count)
        ops.count_valid_labels(labels, e.C, self.count)
        self._sync = reducer is not None
        if reducer is not None:
            reducer.allreduce_now(self.count)
        self._grads, tape = {}, []
        P = lambda scope, kind: self.view(self.param, scope, kind)    # noqa: E731
        G = lambda scope, kind: self.view(self.grad, scope, kind)     # noqa: E731
        wkey = ('wgrad_ws', n, h, w)
        wws = self._a.get(wkey)          # split-K slabs of the filter gradients; sized after the forward pass, read by
        wneed = [0]                      # the backward closures (which run after it is assigned)

        def conv_bn(scope, xname, xact, k, relu, out, key=None, cout=None, scatter=None, bn_scope=None):
            """out = [relu](BN(conv(x))); bn_scope: where gamma / beta live (a list of (scope, channels) for the stacked
            atrous pair, whose batch norm is two independent ones side by side)."""
            key = key or scope
            cout = cout or self.units[scope][2]
            has_bias = key == scope and self.units[scope][3]
            bias = P(scope, 'bias') if has_bias else self.zeros[:cout]
            z = ops.conv2d_fwd(xact, self.w[key], bias, k, relu=False, y=self._act('z_' + out, xact.n, xact.h, xact.w, cout))[0]
            wneed[0] = max(wneed[0], ops.conv2d_bwd_filter_workspace_bytes(xact, cout, k))
            y, bwd_bn = self._bn_pair(bn_scope, z, relu, out) if bn_scope else self._bn_unit(scope, z, relu, out)

            def bwd():
                dy = self._grads.pop(out)[0]
                dz = bwd_bn(dy)
                if scatter is None:
                    ops.conv2d_bwd_filter(xact, dz, G(scope, 'kernel'), G(scope, 'bias') if has_bias else None, k,
                                          workspace=wws)
                else:
                    dw = self._scratch.get(('dw', key))
                    if dw is None:
                        dw = self._scratch[('dw', key)] = torch.empty(k, k, xact.c, cout, dtype=torch.float32,
                                                                      device=e.device)
                    ops.zero_(dw)
                    ops.conv2d_bwd_filter(xact, dz, dw, G(scope, 'bias') if has_bias else None, k, workspace=wws)
                    scatter(dw)
                if xname is not None:
                    prev = self._grads.get(xname) if _FUSE_ACCUM else None
                    if prev is not None:
                        # a gradient of this tensor already exists (the other branch of a residual block): the data-gradient
                        # conv adds it in its own epilogue -- fp32 sum, one rounding -- instead of an add pass over three maps
                        dx = ops.conv2d_bwd_data(dz, self.wd[key], self.zeros[:xact.c], self._like('dx_' + out, xact), k,
                                                 addend=prev[0])
                        self._grads[xname] = (dx, False)
                    else:
                        dx = ops.conv2d_bwd_data(dz, self.wd[key], self.zeros[:xact.c], self._like('dx_' + out, xact), k)
                        self._accum(xname, dx)
            tape.append(bwd)
            return y

        # ---- forward -----------------------------------------------------------------------------------------------
        z0 = self._act('z_block_0_1', n, h, w, 64)
        ops.conv2d_first_fwd(x.contiguous(), P('block_0_1', 'kernel'), P('block_0_1', 'bias'), z0, relu=False)
        y0, bwd_bn0 = self._bn_unit('block_0_1', z0, True, 'block_0_1')

        def bwd_first():
            dz = bwd_bn0(self._grads.pop('block_0_1')[0])
            ops.conv2d_first_bwd_filter(x, dz, G('block_0_1', 'kernel'), G('block_0_1', 'bias'))
        tape.append(bwd_first)

        g7 = ops.gather_conv7s2(y0, self._act('op_block_0_2', n, h // 2, w // 2, 576))

        def bwd_gather7():
            self._accum('block_0_1', ops.gather_conv7s2_bwd(self._grads.pop('op_block_0_2')[0], self._like('dg7', y0)))
        tape.append(bwd_gather7)

        def scatter7(dw):
            G('block_0_2', 'kernel').view(-1).add_(dw.view(-1)[self.inv7])
        y02 = conv_bn('block_0_2', 'op_block_0_2', g7, 3, True, 'block_0_2', scatter=scatter7)
        cur = ops.maxpool2x2_fwd(y02, self._act('block_0_pool', n, h // 4, w // 4, 64))

        def bwd_pool():
            self._accum('block_0_2', ops.maxpool2x2_bwd(y02, self._grads.pop('block_0_pool')[0], self._like('dpool', y02)))
        tape.append(bwd_pool)
        curname = 'block_0_pool'

        def residual(a_name, a, b_name, b, out):
            y = ops.add(a, b, self._like(out, a))

            def bwd():
                dy = self._grads.pop(out)[0]
                self._accum(a_name, dy)
                self._accum(b_name, dy)
            tape.append(bwd)
            return y

        shortcut = None
        for index, (name, kind, args) in enumerate(self.blocks, start=1):
            inp, inpname = cur, curname
            if kind == 'a':
                mid, cout, stride, shortcut_conv = args
                if stride == 2:
                    full, fullname = cur, curname
                    inp = ops.subsample2(cur, self._act(name + '/input', cur.n, cur.h // 2, cur.w // 2, cur.c))
                    inpname = name + '/input'

                    def bwd_sub(full=full, fullname=fullname, inpname=inpname):
                        self._accum(fullname, ops.subsample2_bwd(self._grads.pop(inpname)[0], self._like('dsub_' + inpname, full)))
                    tape.append(bwd_sub)
                s1 = conv_bn(name + '/stage_1', inpname, inp, 1, True, name + '/s1')
                s2 = conv_bn(name + '/stage_2', name + '/s1', s1, 3, True, name + '/s2')
            else:
                f1, f2, cout, d1, d2, shortcut_conv = args
                s1 = conv_bn(name + '/stage_1', inpname, inp, 1, True, name + '/s1')
                pair_bn = [(name + '/stage_2_1', f2 // 2), (name + '/stage_2_2', f2 // 2)]
                if name in self.wd_pair and ops.dilated_pair_training_ok(s1, f2):
                    # the pair forward and backward WITHOUT the 18 f1 operand: the taps gathered by the GEMMs' own loads, only
                    # the diagonal blocks of the block-diagonal product computed (ops.conv_dilated_pair*)
                    z2 = ops.conv_dilated_pair(s1, self.w[name + '/stage_2'], self.zeros[:f2], d1, d2, relu=False,
                                               y=self._act('z_' + name + '/s2', s1.n, s1.h, s1.w, f2))
                    wneed[0] = max(wneed[0], ops.conv_dilated_pair_bwd_filter_workspace_bytes(s1, f2))
                    s2, bwd_bn2 = self._bn_pair(pair_bn, z2, True, name + '/s2')

                    def bwd_pair(name=name, s1=s1, d1=d1, d2=d2, bwd_bn2=bwd_bn2):
                        dz = bwd_bn2(self._grads.pop(name + '/s2')[0])
                        ops.conv_dilated_pair_bwd_filter(s1, dz, d1, d2, G(name + '/stage_2_1', 'kernel'),
                                                         G(name + '/stage_2_2', 'kernel'), wws)
                        self._accum(name + '/s1', ops.conv_dilated_pair_bwd_data(dz, self.wd_pair[name], self.zeros[:s1.c], d1, d2,
                                                                                 self._like('dx_' + name + '/s2', s1)))
                    tape.append(bwd_pair)
                else:
                    op = ops.im2col_dilated_pair(s1, d1, d2, self._act(name + '/operand', s1.n, s1.h, s1.w, 18 * f1))

                    def bwd_col(name=name, s1=s1, d1=d1, d2=d2):
                        self._accum(name + '/s1', ops.im2col_dilated_pair_bwd(self._grads.pop(name + '/operand')[0], d1, d2,
                                                                              self._like('dcol_' + name, s1)))
                    tape.append(bwd_col)

                    def scatter_pair(dw, name=name, f1=f1, f2=f2):
                        half = f2 // 2
                        G(name + '/stage_2_1', 'kernel').add_(dw[0, 0, :9 * f1, :half].reshape(3, 3, f1, half))
                        G(name + '/stage_2_2', 'kernel').add_(dw[0, 0, 9 * f1:, half:].reshape(3, 3, f1, half))
                    s2 = conv_bn(name + '/stage_2_1', name + '/operand', op, 1, True, name + '/s2', key=name + '/stage_2',
                                 cout=f2, scatter=scatter_pair, bn_scope=pair_bn)
            s3 = conv_bn(name + '/stage_3', name + '/s2', s2, 1, True, name + '/s3')
            if shortcut_conv:
                short, shortname = conv_bn(name + '/shortcut', inpname, inp, 1, True, name + '/sc'), name + '/sc'
            else:
                short, shortname = inp, inpname
            curname = 'block_%d' % index
            cur = residual(name + '/s3', s3, shortname, short, curname)
            if name == 'block_layer_7':
                shortcut = conv_bn('shortcut', curname, cur, 1, False, 'shortcut')
        d = conv_bn('first_deconvolution_conv', curname, cur, 1, True, 'deconv_in')
        fu, su = 'first_deconvolution_upconv', 'second_deconvolution_upconv'

        def deconv_grads(scope, stride, xact, dph, xname, dxtag):
            """filter gradient of the dense deconv (through the inverse index map) and its data gradient into `xname`"""
            _, _, inv, kshape = self.dmap[scope]
            dk = self._scratch.get(('dk', scope))
            if dk is None:
                dk = self._scratch[('dk', scope)] = torch.empty(kshape, dtype=torch.float32, device=e.device)
            ops.zero_(dk)
            ops.conv2d_bwd_filter(xact, dph, dk, None, 3, workspace=wws)
            G(scope, 'kernel').view(-1).add_(dk.view(-1)[inv])
            self._accum(xname, ops.conv2d_bwd_data(dph, self.wd[scope], self.zeros[:xact.c], self._like(dxtag, xact), 3))

        # x2 transposed conv [4,4,U,width] as a 3x3 conv onto 4 phases x U channels + depth-to-space
        z_up, self._a['dd_ws1'] = ops.deconv_dense_fwd(d, self.w[fu], self.zeros[:4 * e.Up], 2, e.Up,
                                                       y=self._act('z_deconv_1', d.n, 2 * d.h, 2 * d.w, e.Up), relu=False,
                                                       workspace=self._a.get('dd_ws1'))
        wneed[0] = max(wneed[0], ops.conv2d_bwd_filter_workspace_bytes(d, 4 * e.Up, 3))
        y_up, bwd_bn_up = self._bn_unit(fu, z_up, False, 'deconv_1')

        def bwd_up2():
            dz = bwd_bn_up(self._grads.pop('deconv_1')[0])
            dph = ops.space_to_depth(dz, 2, self._act('dph_deconv_1', d.n, d.h, d.w, 4 * e.Up))
            deconv_grads(fu, 2, d, dph, 'deconv_in', 'd_deconv_in')
        tape.append(bwd_up2)
        merge = residual('deconv_1', y_up, 'shortcut', shortcut, 'merge')
        # ---- head: x8 transposed conv [16,16,C,U] (3x3 conv onto 64 phases x Cp classes at 1/8 resolution, shuffled
        # straight into dense float32 scores), batch norm, softmax cross-entropy ---------------------------------------
        # (float32: the reference computes these scores in float32, adapnet.py:155-163 -- through a bf16 phase map they were
        # rounded to 8 bits in front of the batch norm and the softmax; the backward pass keeps the bf16 MFMA convs)
        raw = ops.deconv8_scores_f32(merge, self.w32_scores, e.C, self.Cp, self._dense('score_raw', (n, h, w, e.C)), self._a)
        wneed[0] = max(wneed[0], ops.conv2d_bwd_filter_workspace_bytes(merge, 64 * self.Cp, 3))
        mm, mv = self.moving[su]
        logits = ops.bn_dense_forward(raw, P(su, 'gamma'), P(su, 'beta'), mm, mv, self.bn[su],
                                      self._dense('logits', (n, h, w, e.C)), sync=self._sync)
        dlogits = ops.softmax_ce_dense(logits, labels, self.count, e.C, self.loss, self._dense('dlogits', (n, h, w, e.C)))
        # ---- backward ------------------------------------------------------------------------------------------------
        if wws is None or wws.numel() * 4 < wneed[0]:
            wws = self._a[wkey] = torch.empty(max(wneed[0] // 4, 1), dtype=torch.float32, device=e.device)
        dscore = ops.bn_dense_backward(dlogits, raw, P(su, 'gamma'), self.bn[su], G(su, 'gamma'), G(su, 'beta'),
                                       self._dense('dscore', (n, h, w, e.C)), sync=self._sync)
        dph8 = ops.space_to_depth_dense(dscore, 8, self._act('dph_score', merge.n, merge.h, merge.w, 64 * self.Cp))
        deconv_grads(su, 8, merge, dph8, 'merge', 'd_merge')
        for bwd in reversed(tape):
            bwd()
        if reducer is not None:
            reducer.launch(self.grad, (0, self.total))
            reducer.wait()
        self.t += 1
        count = max(int(self.count.item()), 1)
        self.grad_scale = 1.0 / count                     # the second normalisation of adapnet.py:202-203
        self.loss.mul_(self.grad_scale)
        FcnTrainer._apply(self, self.grad_scale)
        self.repack()
        return self.loss

    def _bn_pair(self, scopes, z, relu, name):
        """Batch norm of the stacked atrous pair: per-channel, so one pass over the concatenated map with gamma / beta of
        the two scopes side by side; the moving statistics and parameter gradients are split back afterwards."""
        (s1, c1), (s2, c2) = scopes
        key = ('pairbn', name)
        st = self._scratch.get(key)
        if st is None:
            dev = self.e.device
            st = self._scratch[key] = {'bn': ops.BnState(c1 + c2, dev),
                                       'gamma': torch.empty(c1 + c2, device=dev), 'beta': torch.empty(c1 + c2, device=dev),
                                       'mm': torch.empty(c1 + c2, device=dev), 'mv': torch.empty(c1 + c2, device=dev),
                                       'dg': torch.empty(c1 + c2, device=dev), 'db': torch.empty(c1 + c2, device=dev)}
        P = lambda s, kind: self.view(self.param, s, kind)     # noqa: E731
        torch.cat([P(s1, 'gamma'), P(s2, 'gamma')], out=st['gamma'])
        torch.cat([P(s1, 'beta'), P(s2, 'beta')], out=st['beta'])
        torch.cat([self.moving[s1][0], self.moving[s2][0]], out=st['mm'])
        torch.cat([self.moving[s1][1], self.moving[s2][1]], out=st['mv'])
        y = ops.bn_forward(z, st['gamma'], st['beta'], st['mm'], st['mv'], st['bn'], self._like('y_' + name, z), relu=relu,
                           sync=self._sync)
        self.moving[s1][0].copy_(st['mm'][:c1])
        self.moving[s2][0].copy_(st['mm'][c1:])
        self.moving[s1][1].copy_(st['mv'][:c1])
        self.moving[s2][1].copy_(st['mv'][c1:])

        def bwd_bn(dy):
            ops.zero_(st['dg'])
            ops.zero_(st['db'])
            dz = ops.bn_backward(dy, y if relu else None, z, st['gamma'], st['bn'], st['dg'], st['db'],
                                 self._like('dz_' + name, z), sync=self._sync)
            self.view(self.grad, s1, 'gamma').add_(st['dg'][:c1])
            self.view(self.grad, s2, 'gamma').add_(st['dg'][c1:])
            self.view(self.grad, s1, 'beta').add_(st['db'][:c1])
            self.view(self.grad, s2, 'beta').add_(st['db'][c1:])
            return dz
        return y, bwd_bn
