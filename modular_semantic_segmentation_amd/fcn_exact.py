"""The FCN expert in plain float32 (`conv_dtype='fp32'`, "exact" mode): the same graph as fcn.FcnEngine
(xview/models/simple_fcn.py:10-170) on dense unpadded NHWC float32 maps through csrc/exact_f32.hip -- the reference
graph's own arithmetic type, no bf16 storage.  The convs run on the fp32 matrix instruction (v_mfma_f32_32x32x2_f32: an
fp32 fmaf chain bit for bit, 1/16 of the bf16 MFMA rate), the 2x2 max-pools leave the conv launches.  It is the label-exact
product mode (tests/test_exact_f32_gpu.py: label maps equal to the fp32 oracle's on trained weights), about 1/9 of the bf16
path's images/s.  Same surface as FcnEngine where the fusion models need it: load / encoder / lowres_scores / forward."""
import ctypes
import os

import numpy as np
import torch

from . import _lib, ops
from .custom_layers import is_bilinear_filter
from .fcn import BN_EPS, ENCODER, _fold_bn, variable_shapes


# XV_EXACT_SCALAR=1: round 4's vector-ALU conv kernel + stand-alone pools (A/B baseline of the bench record)
SCALAR_KERNEL = os.environ.get('XV_EXACT_SCALAR', '0') == '1'


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


class FcnEngineF32(object):
    conv_dtype = 'fp32'

    def __init__(self, prefix, in_channels, num_units, num_classes, variables, device='cuda', conv_dtype='fp32', **unused):
        if conv_dtype != 'fp32':
            raise ValueError("FcnEngineF32 is the conv_dtype='fp32' engine")
        self.prefix, self.cin, self.U, self.C = prefix, int(in_channels), int(num_units), int(num_classes)
        self.device = torch.device(device)
        self.fp8_scales = None
        self._arena = {}
        self.load(variables)

    def load(self, variables):
        p, dev = self.prefix, self.device
        v = {k: np.asarray(a, np.float32) for k, a in variables.items() if k.startswith(p + '/')}
        for need, shape in variable_shapes(p, self.cin, self.U, self.C).items():
            if need not in v:
                raise KeyError('missing variable %s' % need)
            if tuple(v[need].shape) != tuple(shape):
                raise ValueError('variable %s has shape %s, expected %s' % (need, v[need].shape, shape))
        def up(a):
            return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)

        # batch norms after the deconvs (custom_layers.py:112-119: BN before the relu; the reference's shipped configuration):
        # a per-channel affine in the x2 kernel / the un-commuted float32 head.  Trained (non-bilinear) deconv kernels: not here.
        self.affine = {}
        for name in ('upscore_conv5', 'upscore'):
            layer = '%s/%s' % (p, name)
            if not is_bilinear_filter(v[layer + '/kernel']):
                raise NotImplementedError("conv_dtype='fp32' evaluates the constant bilinear deconvs (%s is something else): "
                                          'use the bf16 engine' % layer)
            if layer + '/gamma' in v:
                sc = v[layer + '/gamma'] / np.sqrt(v[layer + '/moving_variance'] + BN_EPS)
                self.affine[name] = (up(sc), up(v[layer + '/beta'] - v[layer + '/moving_mean'] * sc))

        self.w, self.b = {}, {}
        for name in [n for n, _, _ in ENCODER] + ['score_conv4', 'score_conv5', 'score']:
            k, b = _fold_bn(v, '%s/%s' % (p, name), v['%s/%s/kernel' % (p, name)], v['%s/%s/bias' % (p, name)])
            self.w[name], self.b[name] = up(k), up(b)
        self.w['score'] = self.w['score'].reshape(self.U, self.C).contiguous()
        torch.cuda.synchronize(dev)

    # ---- ops ---------------------------------------------------------------------------------------------------------
    def _buf(self, name, shape):
        key = (name,) + tuple(shape)
        t = self._arena.get(key)
        if t is None:
            t = self._arena[key] = torch.zeros(shape, dtype=torch.float32, device=self.device)
        return t

    def _conv(self, name, x, k, relu=True, pool=None):
        """conv (+ bias, relu) -> full map; with `pool` also the 2x2 max-pooled map, from the same launch."""
        n, h, w, cin = x.shape
        cout = self.b[name].numel()
        y = self._buf(name, (n, h, w, cout))
        q = self._buf(pool, (n, h // 2, w // 2, cout)) if pool else None
        with ops._Profiled('k3f32' if k == 3 else 'k1f32', 2.0 * n * h * w * cin * cout * k * k):
            if SCALAR_KERNEL:
                rc = _lib.lib().xv_conv2d_f32_scalar(_p(x), n, h, w, cin, _p(self.w[name]), _p(self.b[name]), k, cout, int(relu),
                                                     _p(y), ops._stream())
                if rc == 0 and q is not None:
                    rc = _lib.lib().xv_maxpool2x2_f32(_p(y), n, h, w, cout, _p(q), ops._stream())
            else:
                rc = _lib.lib().xv_conv2d_f32_pool(_p(x), n, h, w, cin, _p(self.w[name]), _p(self.b[name]), k, cout, int(relu),
                                                   _p(y), _p(q), ops._stream())
        _lib.check(rc, 'xv_conv2d_f32')
        return y, q

    def encoder(self, x, keep_all=False):
        """float32 [N,H,W,cin] -> dict of dense float32 NHWC maps with the reference's layer names."""
        n, h, w, cin = x.shape
        if cin != self.cin:
            raise ValueError('expected %d input channels, got %d' % (self.cin, cin))
        if h % 16 or w % 16:
            raise ValueError('H and W must be multiples of 16 (augmentation.py:244-262 crop_multiple)')
        L = {}
        cur = x.contiguous()
        for name, _, pool in ENCODER:
            cur, q = self._conv(name, cur, 3, pool=pool)
            L[name] = cur
            if pool:
                cur = L[pool] = q
        s4 = L['score_conv4'] = self._conv('score_conv4', L['conv4_3'], 1)[0]
        s5 = L['score_conv5'] = self._conv('score_conv5', L['conv5_3'], 1)[0]
        fused = self._buf('fused', tuple(s4.shape))
        n5, h5, w5, c5 = s5.shape
        if 'upscore_conv5' in self.affine:
            sc, sh = self.affine['upscore_conv5']
            _lib.check(_lib.lib().xv_upsample2x_affine_f32(_p(s5), n5, h5, w5, c5, _p(sc), _p(sh), _p(s4), _p(fused), ops._stream()),
                       'xv_upsample2x_affine_f32')
        else:
            _lib.check(_lib.lib().xv_upsample2x_f32(_p(s5), n5, h5, w5, c5, _p(s4), _p(fused), ops._stream()), 'xv_upsample2x_f32')
        L['fused'] = fused
        return L

    def _scores(self, f):
        n, hi, wi, u = f.shape
        cp = (self.C + 3) // 4 * 4
        S = self._buf('lowres_S', (n, hi + 2, wi + 2, cp))          # zero border from the allocation, never written
        _lib.check(_lib.lib().xv_score_lowres_f32(_p(f), n, hi, wi, u, _p(self.w['score']), self.C, _p(S), ops._stream()),
                   'xv_score_lowres_f32')
        return S, (n, hi, wi)

    def lowres_scores(self, x):
        if not self.commuted_head():
            raise NotImplementedError('no low-resolution class scores behind a batch norm with a shift: use forward()')
        return self._scores(self.encoder(x)['fused'])

    def commuted_head(self):
        """Class scores interpolated at 1/8 resolution (no batch-norm shift between the x8 deconv and its relu)?"""
        return 'upscore' not in self.affine

    def forward(self, x, want=('label',), keep_all=False):
        L = self.encoder(x)
        dev = self.device
        out = {}
        if not self.commuted_head():
            f = L['fused']
            n, hi, wi, u = f.shape
            if 'score' in want:
                out['score'] = torch.empty((n, 8 * hi, 8 * wi, self.C), dtype=torch.float32, device=dev)
            if 'prob' in want:
                out['prob'] = torch.empty((n, 8 * hi, 8 * wi, self.C), dtype=torch.float32, device=dev)
            if 'label' in want or 'classification' in want or not out:
                out['label'] = torch.empty((n, 8 * hi, 8 * wi), dtype=torch.int64, device=dev)
            sc, sh = self.affine['upscore']
            rc = _lib.lib().xv_decoder_head_affine_f32(_p(f), n, hi, wi, u, _p(sc), _p(sh), _p(self.w['score']), _p(self.b['score']),
                                                       self.C, _p(out.get('score')), _p(out.get('prob')), _p(out.get('label')),
                                                       ops._stream())
            _lib.check(rc, 'xv_decoder_head_affine_f32')
            if 'label' in out:
                out['classification'] = out['label']
            out['layers'] = L
            return out
        S, (n, hi, wi) = self._scores(L['fused'])
        if 'score' in want:
            out['score'] = torch.empty((n, 8 * hi, 8 * wi, self.C), dtype=torch.float32, device=dev)
        if 'prob' in want:
            out['prob'] = torch.empty((n, 8 * hi, 8 * wi, self.C), dtype=torch.float32, device=dev)
        if 'label' in want or 'classification' in want:
            out['label'] = torch.empty((n, 8 * hi, 8 * wi), dtype=torch.int64, device=dev)
        rc = _lib.lib().xv_decoder_head_from_scores(_p(S), _p(self.b['score']), n, hi, wi, self.C, _p(out.get('score')),
                                                    _p(out.get('prob')), _p(out.get('label')), ops._stream())
        _lib.check(rc, 'xv_decoder_head_from_scores')
        if 'label' in out:
            out['classification'] = out['label']
        out['layers'] = L
        return out

    def calibrate(self, x, margin_bits=1):
        return {}
