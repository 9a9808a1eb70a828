"""Thin torch-tensor wrappers over the C ABI (pointers + sizes in, nothing else).

PyTorch is plumbing here: it owns device memory and the stream; every arithmetic op of the
hot path runs in libxview_hip.so.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import xv_act


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _need(t, dtype, name):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == dtype and t.is_contiguous()):
        raise TypeError('%s must be a contiguous CUDA tensor of dtype %s' % (name, dtype))


BF16, FP8 = 0, 1            # xv_act.dtype (include/xview_hip.h)
FP8_MAX = 448.0             # largest finite OCP e4m3fn


class Act(object):
    """padded-NHWC activation (include/xview_hip.h `xv_act`): dense [n][h+2][w+2][c] with a zero 1-pixel border
    that no kernel ever writes.  dtype 'bf16' (default) or 'fp8' (OCP e4m3fn bytes; a stored q stands for
    q * 2**scale_exp -- the forward convolutions' fp8 path)."""

    def __init__(self, n, h, w, c, device='cuda', dtype='bf16', scale_exp=0):
        self.n, self.h, self.w, self.c = int(n), int(h), int(w), int(c)
        self.dtype, self.scale_exp = dtype, int(scale_exp)
        tdt = torch.bfloat16 if dtype == 'bf16' else torch.float8_e4m3fn
        self.t = torch.zeros((self.n, self.h + 2, self.w + 2, self.c), dtype=tdt, device=device)
        self._xv = xv_act(self.t.data_ptr(), self.n, self.h, self.w, self.c, BF16 if dtype == 'bf16' else FP8,
                          self.scale_exp)

    def xv(self):
        return ctypes.byref(self._xv)

    def set_scale_exp(self, scale_exp):
        self.scale_exp = self._xv.scale_exp = int(scale_exp)

    def interior(self):
        """Logical [n][h][w][c] view (storage dtype)."""
        return self.t[:, 1:-1, 1:-1, :]

    def real(self):
        """Interior as float32 in real units (test / calibration helper)."""
        v = self.interior().float()
        return v * (2.0 ** self.scale_exp) if self.dtype == 'fp8' else v

    @classmethod
    def from_dense(cls, x, dtype='bf16', scale_exp=0):
        """Test helper: pad a dense NHWC float tensor into a new Act (rounds to the storage dtype; fp8 stores
        x / 2**scale_exp, saturating)."""
        n, h, w, c = x.shape
        a = cls(n, h, w, c, device=x.device, dtype=dtype, scale_exp=scale_exp)
        if dtype == 'bf16':
            a.interior().copy_(x.to(torch.bfloat16))
        else:
            a.interior().copy_((x.float() * (2.0 ** -scale_exp)).clamp(-FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn))
        return a


def fp8_scale_exp(amax, margin_bits=0):
    """Smallest power-of-two exponent e with amax / 2**e <= 448 (plus `margin_bits` of headroom)."""
    import math
    amax = float(amax)
    if not amax > 0:
        return 0
    return int(math.ceil(math.log2(amax / FP8_MAX))) + int(margin_bits)


def fp8_scale_exp_mse(values, margin_bits=0, search=6, max_elems=1 << 24):
    """The power-of-two exponent e that minimises the squared error of `values` (a device tensor) stored as e4m3 of
    value / 2**e, saturating at +-448: between the exponent that fits the largest magnitude (+ margin_bits) and `search`
    below it.  A map whose magnitudes have a long tail (the depth expert: raw uint16 depth, logit scale 360) keeps three
    mantissa bits for its bulk instead of spending the format's range on a few outliers, which then saturate.  Calibration
    time only (torch ops on a strided sample of at most max_elems values)."""
    v = values.reshape(-1)
    if v.numel() > max_elems:
        v = v[::(v.numel() + max_elems - 1) // max_elems]
    v = v.float()
    amax = float(v.abs().max())
    if not amax > 0:
        return 0
    top = fp8_scale_exp(amax, margin_bits)
    best, best_err = top, None
    for e in range(top, top - search - 1, -1):
        q = (v * (2.0 ** -e)).clamp_(-FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn).float() * (2.0 ** e)
        err = float(((q - v) ** 2).sum())
        if best_err is None or err < best_err:
            best, best_err = e, err
    return best


_NULL_ACT = ctypes.POINTER(xv_act)()

# bench.py sets this to a list to collect (kind, flops, start_event, end_event) per MFMA-conv launch;
# the events are recorded on the stream the kernel is launched on.
CONV_PROFILE = None


def packed_weight_elems(k, cin, cout):
    """bf16 elements of the packed weight buffer of a k x k conv (3x3: three images, see xview_hip.h)."""
    return _lib.lib().xv_packed_weight_bytes(k, cin, cout) // 2


def pack_conv_weights(w_hwio):
    """float32 HWIO device tensor -> packed bf16 weight buffer for conv2d_fwd."""
    _need(w_hwio, torch.float32, 'w_hwio')
    k, k2, cin, cout = w_hwio.shape
    nbytes = _lib.lib().xv_packed_weight_bytes(k, cin, cout)
    if k != k2 or nbytes == 0:
        raise _lib.XvError('unsupported conv weight shape %s' % (tuple(w_hwio.shape),))
    out = torch.empty(nbytes // 2, dtype=torch.bfloat16, device=w_hwio.device)
    _lib.check(_lib.lib().xv_pack_conv_weights(_ptr(w_hwio), _ptr(out), k, cin, cout, _stream()), 'xv_pack_conv_weights')
    return out


def pack_conv_weights_f8(w_hwio, scale_exp=None):
    """float32 HWIO device tensor -> (packed e4m3 weight buffer with its scale-exponent header, exponent); the
    exponent defaults to the smallest one that keeps max|w| finite."""
    _need(w_hwio, torch.float32, 'w_hwio')
    k, k2, cin, cout = w_hwio.shape
    nbytes = _lib.lib().xv_packed_weight_bytes_f8(k, cin, cout)
    if k != k2 or nbytes == 0:
        raise _lib.XvError('unsupported fp8 conv weight shape %s' % (tuple(w_hwio.shape),))
    if scale_exp is None:
        scale_exp = fp8_scale_exp(w_hwio.abs().max().item())        # one-off, at load time
    out = torch.empty(nbytes, dtype=torch.uint8, device=w_hwio.device)
    _lib.check(_lib.lib().xv_pack_conv_weights_f8(_ptr(w_hwio), _ptr(out), k, cin, cout, int(scale_exp), _stream()),
               'xv_pack_conv_weights_f8')
    return out, int(scale_exp)


def streamk_workspace(device):
    """A zeroed stream-K workspace (xv_conv2d_streamk_workspace_bytes): hand it to every conv2d_fwd / conv2d_bwd_data of
    ONE stream; two streams (the two experts of a fusion model) need one each."""
    return torch.zeros(_lib.lib().xv_conv2d_streamk_workspace_bytes(), dtype=torch.uint8, device=device)


def split_workspace(x, cout, arena, key='split_ws'):
    """The fp32 slab workspace of the split form for a 3x3 conv of Act x onto `cout` channels (xv_conv2d_split_workspace_bytes),
    kept in the caller's `arena` dict under `key` and grown as needed; None where this shape / batch is never split.  One per
    stream: the launch owns it until it has completed."""
    need = _lib.lib().xv_conv2d_split_workspace_bytes(x.n, x.h, x.w, x.c, cout)
    if need == 0:
        return None
    ws = arena.get(key)
    if ws is None or ws.numel() * 4 < need:
        ws = arena[key] = torch.empty((need + 3) // 4, dtype=torch.float32, device=x.t.device)
    return ws


def conv2d_fwd(x, w_packed, bias, k, relu=True, y=None, pooled=None, write_y=True, cfg=-1, workspace=None, split_ws=None):
    """x: Act; returns (y Act or None, pooled Act or None).  An fp8 `x` needs weights from pack_conv_weights_f8; the
    dtype / scale of y and pooled (which must agree) select the output conversion.  workspace (streamk_workspace): the
    generation-2 kernel deals the items of an incomplete last round of tiles out over all CUs (stream-K tail).  split_ws
    (split_workspace): layers whose tiles fill less than half the CUs run in the split form of the 24x16-tile kernel (batch-1
    latency; the same bits as the unsplit launch)."""
    cout = bias.numel()
    _need(bias, torch.float32, 'bias')
    if y is None and write_y:
        y = Act(x.n, x.h, x.w, cout, x.t.device)
    if y is not None:
        ydesc = y._xv
    else:   # pooled-only: the descriptor still names the output dtype
        ydesc = xv_act(None, x.n, x.h, x.w, cout, pooled._xv.dtype, pooled._xv.scale_exp)
    prof = CONV_PROFILE
    if prof is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    if workspace is None and split_ws is not None:
        rc = _lib.lib().xv_conv2d_fwd_split(x.xv(), _ptr(w_packed), _ptr(bias), ctypes.byref(ydesc),
                                           pooled.xv() if pooled is not None else _NULL_ACT, k, int(bool(relu)), int(cfg),
                                           _ptr(split_ws), split_ws.numel() * split_ws.element_size(), _stream())
    elif workspace is None:
        rc = _lib.lib().xv_conv2d_fwd_cfg(x.xv(), _ptr(w_packed), _ptr(bias), ctypes.byref(ydesc),
                                         pooled.xv() if pooled is not None else _NULL_ACT, k, int(bool(relu)), int(cfg),
                                         _stream())
    else:
        rc = _lib.lib().xv_conv2d_fwd_ws(x.xv(), _ptr(w_packed), _ptr(bias), ctypes.byref(ydesc),
                                        pooled.xv() if pooled is not None else _NULL_ACT, k, int(bool(relu)), int(cfg),
                                        _ptr(workspace), workspace.numel() * workspace.element_size(), _stream())
    _lib.check(rc, 'xv_conv2d_fwd')
    if prof is not None:
        ev1.record()
        prof.append(('k%d%s' % (k, 'f8' if x.dtype == 'fp8' else ''),
                     2.0 * x.n * x.h * x.w * x.c * cout * k * k, ev0, ev1))
    return y, pooled


def conv2d_fwd_pair(xa, wa, ba, xb, wb, bb, relu=True, ya=None, yb=None, pa=None, pb=None):
    """The same 3x3 layer of two models in ONE launch (xv_conv2d_fwd_pair): per model the arguments of conv2d_fwd; the
    outputs wanted (full maps ya / yb, pooled maps pa / pb) must be given and be the same set for both.  Returns False --
    nothing launched -- where the shape does not run on the generation-4 / 5 kernels: the caller launches two conv2d_fwd."""
    _need(ba, torch.float32, 'bias_a')
    _need(bb, torch.float32, 'bias_b')
    cout = ba.numel()
    if xa.dtype != 'bf16' or xb.dtype != 'bf16' or (ya is None) != (yb is None) or (pa is None) != (pb is None) or \
            (ya is None and pa is None):
        return False

    def desc(y, x, p):
        return y._xv if y is not None else xv_act(None, x.n, x.h, x.w, cout, p._xv.dtype, p._xv.scale_exp)
    da, db = desc(ya, xa, pa), desc(yb, xb, pb)
    prof = CONV_PROFILE
    if prof is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    rc = _lib.lib().xv_conv2d_fwd_pair(xa.xv(), _ptr(wa), _ptr(ba), ctypes.byref(da), pa.xv() if pa is not None else _NULL_ACT,
                                       xb.xv(), _ptr(wb), _ptr(bb), ctypes.byref(db), pb.xv() if pb is not None else _NULL_ACT,
                                       int(bool(relu)), _stream())
    if rc == -2:
        return False
    _lib.check(rc, 'xv_conv2d_fwd_pair')
    if prof is not None:
        ev1.record()
        prof.append(('k3', 2.0 * 2 * xa.n * xa.h * xa.w * xa.c * cout * 9, ev0, ev1))
    return True


def conv2d_first_fwd(x, w_hwio, bias, y, relu=True):
    _need(x, torch.float32, 'x')
    _need(w_hwio, torch.float32, 'w_hwio')
    _need(bias, torch.float32, 'bias')
    n, h, w, cin = x.shape
    rc = _lib.lib().xv_conv2d_first_fwd(_ptr(x), n, h, w, cin, _ptr(w_hwio), _ptr(bias), y.xv(), int(bool(relu)), _stream())
    _lib.check(rc, 'xv_conv2d_first_fwd')
    return y


def conv2d_first_gather7s2_fwd(x, w_hwio, bias, z, relu=True):
    """relu(conv3x3(x) + b) of the raw input written directly as gather_conv7s2's operand z [N,H/2,W/2,576] (z must hold
    zeros where the gather has no source pixel: a fresh Act, or one only ever written by this op / gather_conv7s2)."""
    _need(x, torch.float32, 'x')
    _need(w_hwio, torch.float32, 'w_hwio')
    _need(bias, torch.float32, 'bias')
    n, h, w, cin = x.shape
    rc = _lib.lib().xv_conv2d_first_gather7s2_fwd(_ptr(x), n, h, w, cin, _ptr(w_hwio), _ptr(bias), z.xv(), int(bool(relu)),
                                                  _stream())
    _lib.check(rc, 'xv_conv2d_first_gather7s2_fwd')
    return z


def conv_first_pair_fwd(x, w1_hwio, b1, w2_packed, b2, y=None, pooled=None, relu1=True, relu2=True):
    """conv1_1 + conv1_2 (+ pool) in one launch (xv_conv_first_pair_fwd): x raw float32 NHWC, y / pooled bf16 Acts (either
    may be None).  Returns False -- nothing launched -- where the fused kernel does not apply (maps that do not tile in
    16x32, other channel counts): the caller then runs conv2d_first_fwd + conv2d_fwd."""
    _need(x, torch.float32, 'x')
    _need(w1_hwio, torch.float32, 'w1_hwio')
    n, h, w, cin = x.shape
    if cin not in (1, 3) or h % 16 or w % 32 or (y is None and pooled is None):
        return False
    prof = CONV_PROFILE
    if prof is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    rc = _lib.lib().xv_conv_first_pair_fwd(_ptr(x), n, h, w, cin, _ptr(w1_hwio), _ptr(b1), int(bool(relu1)), _ptr(w2_packed),
                                           _ptr(b2), int(bool(relu2)), y.xv() if y is not None else _NULL_ACT,
                                           pooled.xv() if pooled is not None else _NULL_ACT, _stream())
    if rc == -2:
        return False
    _lib.check(rc, 'xv_conv_first_pair_fwd')
    if prof is not None:
        # bench.py: a kind of its own (not the plain 3x3 conv kernel the roofline record is about), with the algorithmic FLOPs
        # of BOTH layers; conv1_1's recomputation on the halo and its three-way operand split count as time, not as FLOPs
        ev1.record()
        prof.append(('k3pair', 2.0 * n * h * w * 64 * 9 * (cin + 64), ev0, ev1))
    return True


def maxpool2x2_fwd(x, y=None):
    if y is None:
        y = Act(x.n, x.h // 2, x.w // 2, x.c, x.t.device)
    _lib.check(_lib.lib().xv_maxpool2x2_fwd(x.xv(), y.xv(), _stream()), 'xv_maxpool2x2_fwd')
    return y


def upsample2x_relu_add(x, residual=None, y=None, scale=None, shift=None, relu=True):
    """y = act(bilinear_x2(x) [* scale + shift]) [+ residual]; scale / shift: float32 [C] inference batch norm."""
    if y is None:
        y = Act(x.n, 2 * x.h, 2 * x.w, x.c, x.t.device)
    if scale is not None:
        _need(scale, torch.float32, 'scale')
        _need(shift, torch.float32, 'shift')
    rc = _lib.lib().xv_upsample2x_affine_act_add(x.xv(), _ptr(scale), _ptr(shift),
                                                residual.xv() if residual is not None else _NULL_ACT, y.xv(),
                                                int(bool(relu)), _stream())
    _lib.check(rc, 'xv_upsample2x_affine_act_add')
    return y


def deconv_dense_fwd(x, w_phases_packed, zero_bias, stride, cout, y=None, scale=None, shift=None, residual=None,
                     relu=True, workspace=None):
    """y = act(conv2d_transpose(x, W) [* scale + shift]) [+ residual] for an arbitrary kernel with k = 2 * stride
    (w_phases_packed: pack_conv_weights(dense_deconv_as_conv3x3(W, stride))).  Returns (y, workspace)."""
    if y is None:
        y = Act(x.n, x.h * stride, x.w * stride, cout, x.t.device)
    need = _lib.lib().xv_deconv_dense_workspace_bytes(x.n, x.h, x.w, cout, stride)
    if workspace is None or workspace.numel() * 2 < need:
        workspace = torch.zeros(need // 2, dtype=torch.bfloat16, device=x.t.device)
    rc = _lib.lib().xv_deconv_dense_fwd(x.xv(), _ptr(w_phases_packed), _ptr(zero_bias), _ptr(scale), _ptr(shift),
                                       residual.xv() if residual is not None else _NULL_ACT, y.xv(), int(stride),
                                       int(bool(relu)), _ptr(workspace), workspace.numel() * 2, _stream())
    _lib.check(rc, 'xv_deconv_dense_fwd')
    return y, workspace


def conv1x1_residual(x, w_packed, bias, residual, relu=True, y=None):
    """y = act(conv1x1(x) + bias) + residual (the closing conv of a ResNet block)."""
    _need(bias, torch.float32, 'bias')
    if y is None:
        y = Act(x.n, x.h, x.w, bias.numel(), x.t.device)
    rc = _lib.lib().xv_conv2d_fwd_residual(x.xv(), _ptr(w_packed), _ptr(bias), residual.xv(), y.xv(), int(bool(relu)),
                                          _stream())
    _lib.check(rc, 'xv_conv2d_fwd_residual')
    return y


def subsample2(x, y=None):
    """y[i][j] = x[2i][2j]."""
    if y is None:
        y = Act(x.n, x.h // 2, x.w // 2, x.c, x.t.device)
    _lib.check(_lib.lib().xv_subsample2(x.xv(), y.xv(), _stream()), 'xv_subsample2')
    return y


def gather_conv7s2(x, z=None):
    """[N,H/2,W/2,9C] operand that turns a 7x7 stride-2 'same' conv into a 3x3 stride-1 one (see pack_conv7s2)."""
    if z is None:
        z = Act(x.n, x.h // 2, x.w // 2, 9 * x.c, x.t.device)
    _lib.check(_lib.lib().xv_gather_conv7s2(x.xv(), z.xv(), _stream()), 'xv_gather_conv7s2')
    return z


def subsample2_bwd(dy, dx):
    _lib.check(_lib.lib().xv_subsample2_bwd(dy.xv(), dx.xv(), _stream()), 'xv_subsample2_bwd')
    return dx


def gather_conv7s2_bwd(dz, dx):
    _lib.check(_lib.lib().xv_gather_conv7s2_bwd(dz.xv(), dx.xv(), _stream()), 'xv_gather_conv7s2_bwd')
    return dx


def im2col_dilated_pair_bwd(dz, d1, d2, dx):
    _lib.check(_lib.lib().xv_im2col_dilated_pair_bwd(dz.xv(), int(d1), int(d2), dx.xv(), _stream()),
               'xv_im2col_dilated_pair_bwd')
    return dx


def add(a, b, y=None):
    """y = a + b (Acts of one shape)."""
    if y is None:
        y = Act(a.n, a.h, a.w, a.c, a.t.device)
    _lib.check(_lib.lib().xv_add(a.xv(), b.xv(), y.xv(), _stream()), 'xv_add')
    return y


def space_to_depth(g, stride, out=None):
    """[N,s*H,s*W,C] Act -> [N,H,W,s*s*C] Act, phase channel (py*s + px)*C + c (gradient side of a dense deconv)."""
    if out is None:
        out = Act(g.n, g.h // stride, g.w // stride, stride * stride * g.c, g.t.device)
    _lib.check(_lib.lib().xv_space_to_depth(g.xv(), int(stride), out.xv(), _stream()), 'xv_space_to_depth')
    return out


def space_to_depth_dense(g, stride, out):
    """dense float32 [N,s*H,s*W,C] -> Act [N,H,W,s*s*Cp] (Cp = out.c / s^2 >= C, padding channels zero)."""
    _need(g, torch.float32, 'g')
    _lib.check(_lib.lib().xv_space_to_depth_dense(_ptr(g), int(g.shape[-1]), int(stride), out.xv(), _stream()),
               'xv_space_to_depth_dense')
    return out


def depth_to_space_dense(z, stride, num_classes, out, scale=None, shift=None):
    """phase map Act [N,H,W,s*s*Cp] -> dense float32 [N,s*H,s*W,C] [* scale + shift]."""
    _need(out, torch.float32, 'out')
    if scale is not None:
        _need(scale, torch.float32, 'scale')
        _need(shift, torch.float32, 'shift')
    rc = _lib.lib().xv_depth_to_space_dense(z.xv(), int(stride), int(num_classes), _ptr(scale), _ptr(shift), _ptr(out),
                                            _stream())
    _lib.check(rc, 'xv_depth_to_space_dense')
    return out


def deconv8_scores_f32(x, w_hwio_f32, num_classes, cp, out, arena, scale=None, shift=None):
    """AdapNet's trained x8 score deconv in float32: the padded bf16 map `x` as dense float32 (exact), its 3x3 conv onto the 64
    phases x cp classes (w_hwio_f32 = dense_deconv_as_conv3x3(kernel, 8), float32 [3,3,U,64*cp]) on the fp32 matrix
    instruction, the phases unshuffled into the dense float32 scores `out` [N,8H,8W,C] [* scale + shift].  The scores never
    pass through bf16 (adapnet.py:155-163).  arena: dict for the two float32 scratch maps."""
    _need(out, torch.float32, 'out')
    _need(w_hwio_f32, torch.float32, 'w_hwio_f32')
    lib = _lib.lib()
    xd = arena.get(('d8_x', x.n, x.h, x.w, x.c))
    if xd is None:
        xd = arena[('d8_x', x.n, x.h, x.w, x.c)] = torch.empty((x.n, x.h, x.w, x.c), dtype=torch.float32, device=x.t.device)
    ph = arena.get(('d8_ph', x.n, x.h, x.w, cp))
    if ph is None:
        ph = arena[('d8_ph', x.n, x.h, x.w, cp)] = torch.empty((x.n, x.h, x.w, 64 * cp), dtype=torch.float32, device=x.t.device)
        arena['d8_zero_bias'] = torch.zeros(64 * cp, dtype=torch.float32, device=x.t.device)
    _lib.check(lib.xv_act_to_dense_f32(x.xv(), _ptr(xd), _stream()), 'xv_act_to_dense_f32')
    _lib.check(lib.xv_conv2d_f32(_ptr(xd), x.n, x.h, x.w, x.c, _ptr(w_hwio_f32), _ptr(arena['d8_zero_bias']), 3, 64 * cp, 0,
                                 _ptr(ph), _stream()), 'xv_conv2d_f32')
    _lib.check(lib.xv_depth_to_space_dense_f32(_ptr(ph), x.n, x.h, x.w, 8, cp, num_classes, _ptr(scale), _ptr(shift), _ptr(out),
                                               _stream()), 'xv_depth_to_space_dense_f32')
    return out


def im2col_dilated_pair(x, d1, d2, z=None):
    """[N,H,W,18C]: the nine taps at dilation d1 then the nine at d2."""
    if z is None:
        z = Act(x.n, x.h, x.w, 18 * x.c, x.t.device)
    _lib.check(_lib.lib().xv_im2col_dilated_pair(x.xv(), int(d1), int(d2), z.xv(), _stream()), 'xv_im2col_dilated_pair')
    return z


def dilated_pair_implicit_ok(cin, cout):
    """Shapes xv_conv_dilated_pair_fwd takes: 64-channel K steps; each 128-channel output block inside one half, or one
    64-channel tile for both."""
    return cin % 64 == 0 and (cout % 256 == 0 or cout == 64)


def conv_dilated_pair(x, w_packed, bias, d1, d2, relu=True, y=None):
    """concat(atrous3x3(x, d1), atrous3x3(x, d2)) [+ relu] without the 18C operand (adapnet.py:84-88); w_packed / bias as the
    1x1 conv over im2col_dilated_pair(x) takes them (adapnet.dilated_pair_as_1x1).  Same bits as that pair of calls."""
    _need(bias, torch.float32, 'bias')
    if y is None:
        y = Act(x.n, x.h, x.w, bias.numel(), x.t.device)
    _lib.check(_lib.lib().xv_conv_dilated_pair_fwd(x.xv(), _ptr(w_packed), _ptr(bias), int(d1), int(d2), int(bool(relu)),
                                                   y.xv(), _stream()), 'xv_conv_dilated_pair_fwd')
    return y


def dilated_pair_training_ok(x, cout):
    """Shapes both gradient entries of the implicit pair take (xv_conv_dilated_pair_bwd_data / _bwd_filter_ws)."""
    return x.c % 256 == 0 and cout % 256 == 0 and \
        _lib.lib().xv_conv_dilated_pair_bwd_filter_workspace_bytes(x.n, x.h, x.w, x.c, cout) > 0


def dilated_pair_dgrad_kernel(kernel1, kernel2, out=None):
    """float32 [1,1,18 F/2,C] kernel of the pair's data gradient as a 1x1 conv over (conv, tap, output channel of the conv): row
    (h * 9 + t) * F/2 + co = k_h[t][:, co].  kernel1 / kernel2: [3,3,C,F/2] device tensors; `out`: a tensor to fill in place."""
    _, _, c, half = kernel1.shape
    if out is None:
        out = torch.empty(1, 1, 18 * half, c, dtype=torch.float32, device=kernel1.device)
    v = out.view(2, 9, half, c)
    v[0].copy_(kernel1.reshape(9, c, half).transpose(1, 2))
    v[1].copy_(kernel2.reshape(9, c, half).transpose(1, 2))
    return out


def conv_dilated_pair_bwd_data(dy, w_packed_dgrad, zero_bias, d1, d2, dx):
    """dx = gradient of the pair's input (xv_conv_dilated_pair_bwd_data); w_packed_dgrad: pack_conv_weights of
    dilated_pair_dgrad_kernel(k1, k2)."""
    with _Profiled('dgrad_pair', 2.0 * dy.n * dy.h * dy.w * dx.c * dy.c * 9):
        rc = _lib.lib().xv_conv_dilated_pair_bwd_data(dy.xv(), _ptr(w_packed_dgrad), _ptr(zero_bias), int(d1), int(d2), dx.xv(),
                                                      _stream())
    _lib.check(rc, 'xv_conv_dilated_pair_bwd_data')
    return dx


def conv_dilated_pair_bwd_filter_workspace_bytes(x, cout):
    return _lib.lib().xv_conv_dilated_pair_bwd_filter_workspace_bytes(x.n, x.h, x.w, x.c, cout)


def conv_dilated_pair_bwd_filter(x, dy, d1, d2, dw1, dw2, workspace):
    """dw1 / dw2 (+=): HWIO [3,3,C,F/2] gradients of the two atrous kernels, without the im2col operand."""
    _need(dw1, torch.float32, 'dw1')
    _need(dw2, torch.float32, 'dw2')
    with _Profiled('wgrad_pair', 2.0 * x.n * x.h * x.w * x.c * dy.c * 9):
        rc = _lib.lib().xv_conv_dilated_pair_bwd_filter_ws(x.xv(), dy.xv(), int(d1), int(d2), _ptr(dw1), _ptr(dw2), _ptr(workspace),
                                                           workspace.numel() * 4, _stream())
    _lib.check(rc, 'xv_conv_dilated_pair_bwd_filter_ws')


def dropout(x, rate, seed, y=None):
    """tf.layers.dropout(x, rate, training=True): keep with probability 1 - rate, scale by 1 / (1 - rate)."""
    if y is None:
        y = Act(x.n, x.h, x.w, x.c, x.t.device)
    _lib.check(_lib.lib().xv_dropout(x.xv(), y.xv(), float(rate), int(seed) & 0xffffffffffffffff, _stream()), 'xv_dropout')
    return y


def concat_channels(a, b, y=None):
    if y is None:
        y = Act(a.n, a.h, a.w, a.c + b.c, a.t.device)
    _lib.check(_lib.lib().xv_concat_channels(a.xv(), b.xv(), y.xv(), _stream()), 'xv_concat_channels')
    return y


_HEAD_WS = {}


def _head_workspace(fused, num_classes):
    key = (fused.t.device, fused.n, fused.h, fused.w, num_classes)
    ws = _HEAD_WS.get(key)
    if ws is None:
        nbytes = _lib.lib().xv_decoder_head_workspace_bytes(fused.n, fused.h, fused.w, num_classes)
        ws = torch.empty(nbytes // 4, dtype=torch.float32, device=fused.t.device)
        _HEAD_WS[key] = ws
    return ws


def decoder_head_fwd(fused, w_score, b_score, num_classes, want_score=False, want_prob=False, want_label=True,
                     out=None, workspace=None, scale=None, shift=None):
    """Returns dict with the requested dense outputs (score/prob float32 NHWC, label int64 NHW).
    scale / shift (float32 [U]): batch norm between the x8 deconv and its relu -> the general (un-commuted) head."""
    _need(w_score, torch.float32, 'w_score')
    _need(b_score, torch.float32, 'b_score')
    dev = fused.t.device
    n, ho, wo = fused.n, fused.h * 8, fused.w * 8
    out = {} if out is None else out
    if want_score and 'score' not in out:
        out['score'] = torch.empty((n, ho, wo, num_classes), dtype=torch.float32, device=dev)
    if want_prob and 'prob' not in out:
        out['prob'] = torch.empty((n, ho, wo, num_classes), dtype=torch.float32, device=dev)
    if want_label and 'label' not in out:
        out['label'] = torch.empty((n, ho, wo), dtype=torch.int64, device=dev)
    if scale is not None:
        _need(scale, torch.float32, 'scale')
        _need(shift, torch.float32, 'shift')
        rc = _lib.lib().xv_decoder_head_affine_fwd(fused.xv(), _ptr(scale), _ptr(shift), _ptr(w_score), _ptr(b_score),
                                                  num_classes, _ptr(out.get('score') if want_score else None),
                                                  _ptr(out.get('prob') if want_prob else None),
                                                  _ptr(out.get('label') if want_label else None), _stream())
        _lib.check(rc, 'xv_decoder_head_affine_fwd')
        return out
    ws = workspace if workspace is not None else _head_workspace(fused, num_classes)
    rc = _lib.lib().xv_decoder_head_fwd(fused.xv(), _ptr(w_score), _ptr(b_score), num_classes,
                                       _ptr(out.get('score') if want_score else None),
                                       _ptr(out.get('prob') if want_prob else None),
                                       _ptr(out.get('label') if want_label else None), _ptr(ws), ws.numel() * 4,
                                       _stream())
    _lib.check(rc, 'xv_decoder_head_fwd')
    return out


def score_lowres(fused, w_score, num_classes, S):
    """S[n][i][j][k] = fused . Ws at 1/8 resolution (float32 [N][h+2][w+2][CP], CP = C rounded up to 4)."""
    _lib.check(_lib.lib().xv_score_lowres(fused.xv(), _ptr(w_score), int(num_classes), _ptr(S), _stream()), 'xv_score_lowres')
    return S


def fused_head(Sa, Sb, bias_a, bias_b, n, hi, wi, num_classes, tab, logprior, lognorm=None, out=None):
    """Both experts' low-resolution scores -> the fused label map (int64 [n, 8hi, 8wi]); lognorm given = Dirichlet
    fusion (tab = alpha - 1), else Bayes (tab = log-likelihood tables)."""
    if out is None:
        out = torch.empty((n, 8 * hi, 8 * wi), dtype=torch.int64, device=Sa.device)
    rc = _lib.lib().xv_fused_head_fwd(_ptr(Sa), _ptr(Sb), _ptr(bias_a), _ptr(bias_b), n, hi, wi, int(num_classes),
                                     0 if lognorm is None else 1, _ptr(tab), _ptr(lognorm), _ptr(logprior), _ptr(out),
                                     _stream())
    _lib.check(rc, 'xv_fused_head_fwd')
    return out


def softmax_argmax(score, want_prob=True, want_label=True):
    _need(score, torch.float32, 'score')
    c = score.shape[-1]
    npix = score.numel() // c
    prob = torch.empty_like(score) if want_prob else None
    label = torch.empty(score.shape[:-1], dtype=torch.int64, device=score.device) if want_label else None
    _lib.check(_lib.lib().xv_softmax_argmax(_ptr(score), npix, c, _ptr(prob), _ptr(label), _stream()), 'xv_softmax_argmax')
    return prob, label


def _ptr_array(tensors):
    arr = (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
    return arr


def bayes_fuse(labels, loglik, logprior, want_score=False):
    """labels: list of int64 tensors (same shape); loglik float32 [E,C,C]; logprior float32 [C]."""
    for l in labels:
        _need(l, torch.int64, 'labels')
    _need(loglik, torch.float32, 'loglik')
    _need(logprior, torch.float32, 'logprior')
    e, c = loglik.shape[0], loglik.shape[1]
    npix = labels[0].numel()
    fused = torch.empty_like(labels[0])
    score = torch.empty(tuple(labels[0].shape) + (c,), dtype=torch.float32, device=fused.device) if want_score else None
    rc = _lib.lib().xv_bayes_fuse(_ptr_array(labels), e, _ptr(loglik), _ptr(logprior), c, npix, _ptr(fused), _ptr(score), _stream())
    _lib.check(rc, 'xv_bayes_fuse')
    return fused, score


def bayes_fuse_lut(label_a, label_b, lut):
    _need(label_a, torch.int64, 'label_a')
    _need(label_b, torch.int64, 'label_b')
    _need(lut, torch.int64, 'lut')
    fused = torch.empty_like(label_a)
    rc = _lib.lib().xv_bayes_fuse_lut(_ptr(label_a), _ptr(label_b), _ptr(lut), lut.shape[0], label_a.numel(), _ptr(fused), _stream())
    _lib.check(rc, 'xv_bayes_fuse_lut')
    return fused


def dirichlet_fuse(probs, am1, lognorm, logprior, want_score=False):
    for p in probs:
        _need(p, torch.float32, 'probs')
    e, c = am1.shape[0], am1.shape[1]
    npix = probs[0].numel() // c
    fused = torch.empty(probs[0].shape[:-1], dtype=torch.int64, device=probs[0].device)
    score = torch.empty_like(probs[0]) if want_score else None
    rc = _lib.lib().xv_dirichlet_fuse(_ptr_array(probs), e, _ptr(am1), _ptr(lognorm), _ptr(logprior), c, npix,
                                     _ptr(fused), _ptr(score), _stream())
    _lib.check(rc, 'xv_dirichlet_fuse')
    return fused, score


def average_fuse(probs):
    for p in probs:
        _need(p, torch.float32, 'probs')
    c = probs[0].shape[-1]
    fused = torch.empty(probs[0].shape[:-1], dtype=torch.int64, device=probs[0].device)
    rc = _lib.lib().xv_average_fuse(_ptr_array(probs), len(probs), c, probs[0].numel() // c, _ptr(fused), _stream())
    _lib.check(rc, 'xv_average_fuse')
    return fused


def dirichlet_suffstats(prob, labels, S, counts):
    """Accumulates into S (float64 [C,C]) and counts (int64 [C]) in place."""
    _need(prob, torch.float32, 'prob')
    _need(labels, torch.int32, 'labels')
    _need(S, torch.float64, 'S')
    _need(counts, torch.int64, 'counts')
    c = prob.shape[-1]
    rc = _lib.lib().xv_dirichlet_suffstats(_ptr(prob), _ptr(labels), c, labels.numel(), _ptr(S), _ptr(counts), _stream())
    _lib.check(rc, 'xv_dirichlet_suffstats')


def confusion_matrix(labels, pred, cm):
    """Accumulates into cm (int64 [C,C], rows = ground truth) in place."""
    _need(labels, torch.int32, 'labels')
    _need(pred, torch.int64, 'pred')
    _need(cm, torch.int64, 'cm')
    rc = _lib.lib().xv_confusion_matrix(_ptr(labels), _ptr(pred), cm.shape[0], labels.numel(), _ptr(cm), _stream())
    _lib.check(rc, 'xv_confusion_matrix')


# ---- training ----------------------------------------------------------------------------------------

def pack_conv_weights_dgrad(w_hwio, out=None):
    """float32 HWIO device tensor -> packed bf16 weights of the data-gradient convolution."""
    _need(w_hwio, torch.float32, 'w_hwio')
    k, _, cin, cout = w_hwio.shape
    if out is None:
        out = torch.empty(_lib.lib().xv_packed_weight_bytes(k, cout, cin) // 2, dtype=torch.bfloat16, device=w_hwio.device)
    _lib.check(_lib.lib().xv_pack_conv_weights_dgrad(_ptr(w_hwio), _ptr(out), k, cin, cout, _stream()),
               'xv_pack_conv_weights_dgrad')
    return out


def pack_conv_weights_into(w_hwio, out):
    k, _, cin, cout = w_hwio.shape
    _lib.check(_lib.lib().xv_pack_conv_weights(_ptr(w_hwio), _ptr(out), k, cin, cout, _stream()), 'xv_pack_conv_weights')
    return out


def pack_conv_weights_pair(w_hwio, out, out_dgrad):
    """Forward and data-gradient packed images of one kernel in one launch."""
    _need(w_hwio, torch.float32, 'w_hwio')
    k, _, cin, cout = w_hwio.shape
    _lib.check(_lib.lib().xv_pack_conv_weights_pair(_ptr(w_hwio), _ptr(out), _ptr(out_dgrad), k, cin, cout, _stream()),
               'xv_pack_conv_weights_pair')


class PackTable(object):
    """Descriptor table (in device memory) of every (master kernel, packed forward image, packed data-gradient image)
    triple of a trainer; run() re-packs them all in ONE launch.  The tensors must stay where they are."""

    def __init__(self, entries, device):
        """entries: list of (w_hwio float32 [k,k,cin,cout] view, packed bf16 buffer, packed dgrad bf16 buffer or None)."""
        self.keep = list(entries)
        arr = (_lib.xv_pack_desc * len(entries))()
        for i, (w, out, outd) in enumerate(entries):
            _need(w, torch.float32, 'w_hwio')
            k, _, cin, cout = w.shape
            arr[i] = _lib.xv_pack_desc(w.data_ptr(), out.data_ptr(), outd.data_ptr() if outd is not None else None, k, cin, cout, 0)
        raw = bytes(arr)
        self.table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)
        self.n = len(entries)

    def run(self):
        _lib.check(_lib.lib().xv_pack_conv_weights_multi(_ptr(self.table), self.n, _stream()), 'xv_pack_conv_weights_multi')


def zero_(t):
    """Clear a tensor on the current stream with the library's memset (no framework kernel in the training step)."""
    _lib.check(_lib.lib().xv_memset_zero(_ptr(t), t.numel() * t.element_size(), _stream()), 'xv_memset_zero')
    return t


class _Profiled(object):
    """with _Profiled(kind, flops): ... -- a HIP-event pair on the launch stream when bench.py collects CONV_PROFILE."""

    def __init__(self, kind, flops):
        self.kind, self.flops, self.prof = kind, flops, CONV_PROFILE

    def __enter__(self):
        if self.prof is not None:
            self.ev0, self.ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.ev0.record()

    def __exit__(self, *exc):
        if self.prof is not None:
            self.ev1.record()
            self.prof.append((self.kind, self.flops, self.ev0, self.ev1))


def conv2d_bwd_data(dy, w_packed_dgrad, zero_bias, dx, k, relu_ref=None, addend=None, workspace=None):
    with _Profiled('dgrad_k%d' % k, 2.0 * dx.n * dx.h * dx.w * dx.c * dy.c * k * k):
        if workspace is None:
            rc = _lib.lib().xv_conv2d_bwd_data(dy.xv(), _ptr(w_packed_dgrad), _ptr(zero_bias),
                                              relu_ref.xv() if relu_ref is not None else _NULL_ACT,
                                              addend.xv() if addend is not None else _NULL_ACT, dx.xv(), k, _stream())
        else:
            rc = _lib.lib().xv_conv2d_bwd_data_ws(dy.xv(), _ptr(w_packed_dgrad), _ptr(zero_bias),
                                                 relu_ref.xv() if relu_ref is not None else _NULL_ACT,
                                                 addend.xv() if addend is not None else _NULL_ACT, dx.xv(), k,
                                                 _ptr(workspace), workspace.numel() * workspace.element_size(), _stream())
    _lib.check(rc, 'xv_conv2d_bwd_data')
    return dx


def conv2d_bwd_filter_workspace_bytes(x, cout, k):
    return _lib.lib().xv_conv2d_bwd_filter_workspace_bytes(x.n, x.h, x.w, x.c, cout, k)


def conv2d_bwd_filter(x, dy, dw, dbias, k, workspace=None):
    """workspace (float32 tensor of >= conv2d_bwd_filter_workspace_bytes): every partial sum (the pixel splits of dW and
    of the bias gradient) goes to slabs added in a fixed order -- bitwise reproducible; None: fp32 atomics."""
    _need(dw, torch.float32, 'dw')
    with _Profiled('wgrad_k%d' % k, 2.0 * x.n * x.h * x.w * x.c * dy.c * k * k):
        if workspace is None:
            rc = _lib.lib().xv_conv2d_bwd_filter(x.xv(), dy.xv(), _ptr(dw), _ptr(dbias), k, _stream())
        else:
            rc = _lib.lib().xv_conv2d_bwd_filter_ws(x.xv(), dy.xv(), _ptr(dw), _ptr(dbias), k, _ptr(workspace),
                                                   workspace.numel() * 4, _stream())
    _lib.check(rc, 'xv_conv2d_bwd_filter')


def conv2d_first_bwd_filter(x, dy, dw, dbias=None, workspace=None):
    """workspace (float32 tensor of >= conv2d_first_bwd_filter_workspace_bytes): per-workgroup partial sums + a
    fixed-order reduce (bitwise reproducible; needs dbias); None: fp32 atomics."""
    _need(x, torch.float32, 'x')
    n, h, w, cin = x.shape
    if workspace is None or dbias is None:
        rc = _lib.lib().xv_conv2d_first_bwd_filter(_ptr(x), n, h, w, cin, dy.xv(), _ptr(dw), _ptr(dbias), _stream())
    else:
        rc = _lib.lib().xv_conv2d_first_bwd_filter_ws(_ptr(x), n, h, w, cin, dy.xv(), _ptr(dw), _ptr(dbias), _ptr(workspace),
                                                      workspace.numel() * 4, _stream())
    _lib.check(rc, 'xv_conv2d_first_bwd_filter')


def conv2d_first_bwd_filter_workspace_bytes(x):
    n, h, w, cin = x.shape
    return _lib.lib().xv_conv2d_first_bwd_filter_workspace_bytes(n, h, w, cin)


def maxpool2x2_bwd(y, dpooled, dy):
    _lib.check(_lib.lib().xv_maxpool2x2_bwd(y.xv(), dpooled.xv(), dy.xv(), _stream()), 'xv_maxpool2x2_bwd')
    return dy


ROUTED_POOL = os.environ.get('XV_ROUTED_POOL', '1') != '0'     # 0: full maps + xv_maxpool2x2_bwd everywhere (A/B timing)


def conv2d_fwd_route(x, w_packed, bias, pooled, route):
    """pooled = maxpool2x2(relu(conv3x3(x) + bias)) and its route bytes (uint8 tensor of >= n * h/2 * w/2 * cout elements) in
    one launch, no full map (xv_conv2d_fwd_route).  Returns False -- nothing launched -- where the generation-4 kernel does not
    take the shape: the caller then writes the full map (conv2d_fwd) and routes with maxpool2x2_bwd."""
    # (the data-gradient conv that reads the routes runs at the POOLED size: that must tile in 16x32 pixels too)
    if not ROUTED_POOL or x.dtype != 'bf16' or x.h % 32 or x.w % 64:
        return False
    _need(bias, torch.float32, 'bias')
    _need(route, torch.uint8, 'route')
    with _Profiled('k3', 2.0 * x.n * x.h * x.w * x.c * pooled.c * 9):
        rc = _lib.lib().xv_conv2d_fwd_route(x.xv(), _ptr(w_packed), _ptr(bias), pooled.xv(), _ptr(route), route.numel(), _stream())
    if rc == -2:
        return False
    _lib.check(rc, 'xv_conv2d_fwd_route')
    return True


def conv2d_bwd_data_route(dy, w_packed_dgrad, zero_bias, route, dx):
    """dx (twice dy's size) = MaxPoolGrad + ReluGrad of conv3x3(dy, Wd) through the route bytes of conv2d_fwd_route: the bits
    of conv2d_bwd_data onto a pooled-size map + maxpool2x2_bwd, without either map in memory."""
    _need(route, torch.uint8, 'route')
    with _Profiled('dgrad_k3', 2.0 * dy.n * dy.h * dy.w * dx.c * dy.c * 9):
        rc = _lib.lib().xv_conv2d_bwd_data_route(dy.xv(), _ptr(w_packed_dgrad), _ptr(zero_bias), _ptr(route), route.numel(),
                                                 dx.xv(), _stream())
    _lib.check(rc, 'xv_conv2d_bwd_data_route')
    return dx


def relu_bwd(g, ref, out):
    _lib.check(_lib.lib().xv_relu_bwd(g.xv(), ref.xv(), out.xv(), _stream()), 'xv_relu_bwd')
    return out


def upsample2x_bwd(dfused, s5, ds5):
    _lib.check(_lib.lib().xv_upsample2x_bwd(dfused.xv(), s5.xv(), ds5.xv(), _stream()), 'xv_upsample2x_bwd')
    return ds5


def count_valid_labels(labels, num_classes, count):
    _need(labels, torch.int32, 'labels')
    _need(count, torch.int64, 'count')
    _lib.check(_lib.lib().xv_count_valid_labels(_ptr(labels), num_classes, labels.numel(), _ptr(count), _stream()),
               'xv_count_valid_labels')


def decoder_head_bwd(fused, w_score, b_score, labels, count, num_classes, loss, dw_score, db_score, dfused,
                     workspace=None):
    _need(labels, torch.int32, 'labels')
    _need(loss, torch.float64, 'loss')
    if workspace is None:
        nbytes = _lib.lib().xv_decoder_head_bwd_workspace_bytes(fused.n, fused.h, fused.w, num_classes)
        workspace = torch.empty(nbytes // 4, dtype=torch.float32, device=fused.t.device)
    rc = _lib.lib().xv_decoder_head_bwd(fused.xv(), _ptr(w_score), _ptr(b_score), _ptr(labels), _ptr(count), num_classes,
                                       _ptr(loss), _ptr(dw_score), _ptr(db_score), dfused.xv(), _ptr(workspace),
                                       workspace.numel() * 4, _stream())
    _lib.check(rc, 'xv_decoder_head_bwd')
    return workspace


def adam_step(param, grad, m, v, lr_t, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0):
    _lib.check(_lib.lib().xv_adam_step(_ptr(param), _ptr(grad), _ptr(m), _ptr(v), param.numel(), lr_t, beta1, beta2, eps,
                                      grad_scale, _stream()), 'xv_adam_step')


def rmsprop_step(param, grad, ms, lr, decay=0.9, eps=1e-10, grad_scale=1.0):
    _lib.check(_lib.lib().xv_rmsprop_step(_ptr(param), _ptr(grad), _ptr(ms), param.numel(), lr, decay, eps, grad_scale,
                                         _stream()), 'xv_rmsprop_step')


def adagrad_step(param, grad, accum, lr, grad_scale=1.0):
    _lib.check(_lib.lib().xv_adagrad_step(_ptr(param), _ptr(grad), _ptr(accum), param.numel(), lr, grad_scale, _stream()),
               'xv_adagrad_step')


def bias_grad(dy, dbias):
    _need(dbias, torch.float32, 'dbias')
    _lib.check(_lib.lib().xv_bias_grad(dy.xv(), _ptr(dbias), _stream()), 'xv_bias_grad')


# ---- training-mode batch norm and the un-commuted training head (csrc/batchnorm.hip) -------------------------
BN_EPS, BN_MOMENTUM = 1e-3, 0.99        # [TF1] tf.layers.batch_normalization defaults


class BnState(object):
    """Per-layer buffers of one training-mode batch norm: batch statistics, folded scale / shift, scratch."""

    def __init__(self, channels, device, deterministic=True):
        f = lambda: torch.zeros(channels, dtype=torch.float32, device=device)   # noqa: E731
        self.c = channels
        self.mean, self.invstd, self.scale, self.shift = f(), f(), f(), f()
        self.sums = torch.zeros(2 * channels, dtype=torch.float64, device=device)
        # per-workgroup partial sums of the reductions, added in a fixed order (xv_bn_workspace_bytes): bitwise
        # reproducible statistics and gamma / beta gradients; deterministic=False: f64 atomics in arrival order
        self.ws = torch.empty(_lib.lib().xv_bn_workspace_bytes(channels) // 4, dtype=torch.float32, device=device) \
            if deterministic else None

    def wsp(self):
        return (_ptr(self.ws), self.ws.numel() * 4) if self.ws is not None else (None, 0)


def _sync_sums(st, sync):
    """Sync-BN: sum the per-channel statistics over the data-parallel ranks; returns the factor by which the local
    sample count grows.  Every rank holds the same number of images per step: the models check that the ranks agree on
    `batchsize` when the trainer is created and that every data-parallel step has exactly that many images
    (parallel.require_equal_batchsize)."""
    if not sync:
        return 1
    from .parallel import allreduce_stats_, world
    allreduce_stats_(st.sums)           # on a communicator of its own: never queued behind a gradient bucket in flight
    return world()[1]


def conv2d_fwd_stats(x, w_packed, bias, z, st):
    """z = conv3x3(x) + bias with the batch statistics of z taken in the conv kernel's epilogue -- as PER-WORKGROUP ROWS in
    st.conv_rows (st.conv_rows_n of them), NOT in st.sums: the sums are produced by the next bn_forward(have_stats=True)
    (xv_bn_finalize_from_rows / xv_bn_sums_from_rows), until then st.sums still holds the previous step's values.  Returns
    False (nothing launched) where that kernel does not apply: the caller then runs conv2d_fwd and lets bn_forward take the
    statistics."""
    if os.environ.get('XV_BN_CONV_STATS') == '0':        # A/B timing: the separate statistics pass
        return False
    lib = _lib.lib()
    rows = lib.xv_conv2d_stats_rows()
    if getattr(st, 'conv_rows', None) is None or st.conv_rows.numel() < rows * 2 * st.c:
        st.conv_rows = torch.empty(rows * 2 * st.c, dtype=torch.float32, device=z.t.device)
    rc = lib.xv_conv2d_fwd_stats(x.xv(), _ptr(w_packed), _ptr(bias), z.xv(), _ptr(st.conv_rows), st.conv_rows.numel() * 4,
                                 _stream())
    if rc == -2:            # XV_ESHAPE: the generation-4 kernel does not take this shape
        return False
    _lib.check(rc, 'xv_conv2d_fwd_stats')
    st.conv_rows_n = rows       # bn_forward(have_stats=True) adds the rows up (in the launch that finalises, where it can)
    return True


def bn_forward(z, gamma, beta, moving_mean, moving_var, st, y, relu=True, sync=False, pooled=None, have_stats=False,
               ups8_of=None):
    """y = [relu](BN_batch(z)); updates the moving statistics in place (z, y: Act).  sync: statistics over all
    data-parallel ranks (one all-reduce of 2*C doubles).  pooled (with relu): the 2x2 max-pool of y from the same pass; y
    may then be None (only the pooled map is written).  ups8_of (an Act 8 times smaller, with z None): z IS its bilinear x8
    up-sampling and is recomputed per element instead of read (deterministic workspace only; with sync the sums are all-reduced
    between the statistics pass and the per-channel results, like every other batch norm)."""
    lib = _lib.lib()
    fin = (_ptr(gamma), _ptr(beta), BN_EPS, BN_MOMENTUM, _ptr(moving_mean), _ptr(moving_var), _ptr(st.mean), _ptr(st.invstd),
           _ptr(st.scale), _ptr(st.shift))
    ws = st.wsp()
    if ups8_of is not None:
        if st.ws is None or pooled is not None or have_stats:
            raise ValueError('bn_forward(ups8_of=): statistics with a workspace, no pool')
        if sync:    # data parallel: the sums alone, all-reduced over the ranks, then the per-channel results
            _lib.check(lib.xv_bn_stats_ups8_ws(ups8_of.xv(), _ptr(st.sums), ws[0], ws[1], _stream()), 'xv_bn_stats_ups8_ws')
            mult = _sync_sums(st, sync)
            _lib.check(lib.xv_bn_finalize(_ptr(st.sums), st.c, ups8_of.n * 64 * ups8_of.h * ups8_of.w * mult, *fin, _stream()),
                       'xv_bn_finalize')
        else:
            _lib.check(lib.xv_bn_stats_finalize_ups8_ws(ups8_of.xv(), _ptr(st.sums), ws[0], ws[1], *fin, _stream()),
                       'xv_bn_stats_finalize_ups8_ws')
        if y is None:           # statistics only: the apply pass is fused into what follows (score_dense_fwd_ups8)
            return None
        _lib.check(lib.xv_bn_apply_ups8(ups8_of.xv(), _ptr(st.scale), _ptr(st.shift), int(bool(relu)), y.xv(), _stream()),
                   'xv_bn_apply_ups8')
        return y
    if not sync and have_stats:
        # one process: the row sums of the conv epilogue's partial statistics and the per-channel results in ONE launch
        _lib.check(lib.xv_bn_finalize_from_rows(_ptr(st.conv_rows), st.conv_rows_n, st.c, z.n * z.h * z.w, *fin, _ptr(st.sums),
                                                _stream()), 'xv_bn_finalize_from_rows')
    elif not sync and st.ws is not None:
        _lib.check(lib.xv_bn_stats_finalize_ws(z.xv(), _ptr(st.sums), ws[0], ws[1], *fin, _stream()), 'xv_bn_stats_finalize_ws')
    else:
        if have_stats:
            _lib.check(lib.xv_bn_sums_from_rows(_ptr(st.conv_rows), st.conv_rows_n, 2 * st.c, _ptr(st.sums), _stream()),
                       'xv_bn_sums_from_rows')
        else:
            _lib.check(lib.xv_bn_stats_ws(z.xv(), _ptr(st.sums), *ws, _stream()), 'xv_bn_stats_ws')
        mult = _sync_sums(st, sync)
        _lib.check(lib.xv_bn_finalize(_ptr(st.sums), st.c, z.n * z.h * z.w * mult, *fin, _stream()), 'xv_bn_finalize')
    if pooled is not None:
        if not relu:
            raise ValueError('the fused pool follows the relu')
        _lib.check(lib.xv_bn_apply_pool(z.xv(), _ptr(st.scale), _ptr(st.shift), y.xv() if y is not None else _NULL_ACT,
                                        pooled.xv(), _stream()), 'xv_bn_apply_pool')
        return y
    _lib.check(lib.xv_bn_apply(z.xv(), _ptr(st.scale), _ptr(st.shift), int(bool(relu)), y.xv(), _stream()), 'xv_bn_apply')
    return y


def bn_backward(dy, y, z, gamma, st, dgamma, dbeta, dz, sync=False, mask_from_z=None, relu=None, ups8_of=None):
    """dz from dy (gradient w.r.t. the post-relu output); accumulates the LOCAL dgamma / dbeta (the gradient all-reduce
    sums them over ranks).  relu: is there an activation behind the batch norm (default: `y is not None`).  With one, the relu
    mask is recomputed from z (z * scale + shift > 0 with the scale / shift the forward pass left in `st`) for the trunk's
    channel counts (64 .. 2048, powers of two) -- a third less traffic, and `y` is then NOT read and may be None
    (`relu=True`); other channel counts, and mask_from_z=False, read the mask from y, which must then be given."""
    lib = _lib.lib()
    if ups8_of is not None:     # z = bilinear_x8(ups8_of), recomputed per element (relu behind the batch norm, mask from z)
        _lib.check(lib.xv_bn_bwd_reduce_zmask_ups8(dy.xv(), ups8_of.xv(), _ptr(st.mean), _ptr(st.invstd), _ptr(st.scale),
                                                   _ptr(st.shift), _ptr(st.sums), _ptr(dgamma), _ptr(dbeta), *st.wsp(), _stream()),
                   'xv_bn_bwd_reduce_zmask_ups8')
        mult = _sync_sums(st, sync)
        _lib.check(lib.xv_bn_bwd_apply_zmask_ups8(dy.xv(), ups8_of.xv(), _ptr(st.mean), _ptr(st.invstd), _ptr(st.scale),
                                                  _ptr(st.shift), _ptr(gamma), _ptr(st.sums), dy.n * dy.h * dy.w * mult, dz.xv(),
                                                  _stream()), 'xv_bn_bwd_apply_zmask_ups8')
        return dz
    if relu is None:
        relu = y is not None
    can_zmask = z.c >= 64 and 2048 % z.c == 0
    if mask_from_z is None:
        mask_from_z = relu and can_zmask
    if relu and not mask_from_z and y is None:
        raise ValueError('bn_backward: %d channels take the relu mask from y (none given)' % z.c)
    if not relu:
        y = None
    if mask_from_z and relu:
        _lib.check(lib.xv_bn_bwd_reduce_zmask(dy.xv(), z.xv(), _ptr(st.mean), _ptr(st.invstd), _ptr(st.scale), _ptr(st.shift),
                                              _ptr(st.sums), _ptr(dgamma), _ptr(dbeta), *st.wsp(), _stream()),
                   'xv_bn_bwd_reduce_zmask')
        mult = _sync_sums(st, sync)
        _lib.check(lib.xv_bn_bwd_apply_zmask(dy.xv(), z.xv(), _ptr(st.mean), _ptr(st.invstd), _ptr(st.scale), _ptr(st.shift),
                                             _ptr(gamma), _ptr(st.sums), z.n * z.h * z.w * mult, dz.xv(), _stream()),
                   'xv_bn_bwd_apply_zmask')
        return dz
    yx = y.xv() if y is not None else _NULL_ACT
    _lib.check(lib.xv_bn_bwd_reduce_ws(dy.xv(), yx, z.xv(), _ptr(st.mean), _ptr(st.invstd), _ptr(st.sums), _ptr(dgamma),
                                       _ptr(dbeta), *st.wsp(), _stream()), 'xv_bn_bwd_reduce_ws')
    mult = _sync_sums(st, sync)
    _lib.check(lib.xv_bn_bwd_apply(dy.xv(), yx, z.xv(), _ptr(st.mean), _ptr(st.invstd), _ptr(gamma), _ptr(st.sums),
                                   z.n * z.h * z.w * mult, dz.xv(), _stream()), 'xv_bn_bwd_apply')
    return dz


def bn_pool_backward(dpooled, z, gamma, st, dgamma, dbeta, dz, sync=False):
    """dz of a conv -> batch norm -> relu -> 2x2 max-pool block from the gradient of the POOLED map: MaxPoolGrad, ReluGrad
    and the batch-norm gradient in the two normalisation passes (the routed full-resolution gradient never reaches HBM)."""
    lib = _lib.lib()
    _lib.check(lib.xv_bn_pool_bwd_reduce(dpooled.xv(), z.xv(), _ptr(st.mean), _ptr(st.invstd), _ptr(st.scale), _ptr(st.shift),
                                         _ptr(st.sums), _ptr(dgamma), _ptr(dbeta), *st.wsp(), _stream()), 'xv_bn_pool_bwd_reduce')
    mult = _sync_sums(st, sync)
    _lib.check(lib.xv_bn_pool_bwd_apply(dpooled.xv(), z.xv(), _ptr(st.mean), _ptr(st.invstd), _ptr(st.scale), _ptr(st.shift),
                                        _ptr(gamma), _ptr(st.sums), z.n * z.h * z.w * mult, dz.xv(), _stream()),
               'xv_bn_pool_bwd_apply')
    return dz


def bn_dense_forward(z, gamma, beta, moving_mean, moving_var, st, y, sync=False):
    """The same on a dense float32 [..., C] tensor (no activation)."""
    lib = _lib.lib()
    c = z.shape[-1]
    rows = z.numel() // c
    _lib.check(lib.xv_bn_dense_stats_ws(_ptr(z), rows, c, _ptr(st.sums), *st.wsp(), _stream()), 'xv_bn_dense_stats_ws')
    mult = _sync_sums(st, sync)
    _lib.check(lib.xv_bn_finalize(_ptr(st.sums), c, rows * mult, _ptr(gamma), _ptr(beta), BN_EPS, BN_MOMENTUM,
                                  _ptr(moving_mean), _ptr(moving_var), _ptr(st.mean), _ptr(st.invstd), _ptr(st.scale),
                                  _ptr(st.shift), _stream()), 'xv_bn_finalize')
    if y is not None:       # (y = None: the caller applies st.scale / st.shift itself, e.g. softmax_ce_dense(affine=st))
        _lib.check(lib.xv_bn_dense_apply(_ptr(z), rows, c, _ptr(st.scale), _ptr(st.shift), _ptr(y), _stream()),
                   'xv_bn_dense_apply')
    return y


def bn_dense_backward(dy, z, gamma, st, dgamma, dbeta, dz, sync=False):
    lib = _lib.lib()
    c = z.shape[-1]
    rows = z.numel() // c
    _lib.check(lib.xv_bn_dense_bwd_reduce_ws(_ptr(dy), _ptr(z), rows, c, _ptr(st.mean), _ptr(st.invstd), _ptr(st.sums),
                                             _ptr(dgamma), _ptr(dbeta), *st.wsp(), _stream()), 'xv_bn_dense_bwd_reduce_ws')
    mult = _sync_sums(st, sync)
    _lib.check(lib.xv_bn_dense_bwd_apply(_ptr(dy), _ptr(z), rows, c, _ptr(st.mean), _ptr(st.invstd), _ptr(gamma),
                                         _ptr(st.sums), rows * mult, _ptr(dz), _stream()), 'xv_bn_dense_bwd_apply')
    return dz


def upsample_raw_fwd(x, factor, y=None):
    if y is None:
        y = Act(x.n, factor * x.h, factor * x.w, x.c, x.t.device)
    _lib.check(_lib.lib().xv_upsample_raw_fwd(x.xv(), factor, y.xv(), _stream()), 'xv_upsample_raw_fwd')
    return y


_UPS_WS = {}
UPS8_BLOCK_SUMS = os.environ.get('XV_UPS8_BLOCK_SUMS', '1') != '0'      # 0: the gather form of the x8 gradient (A/B timing)


def upsample_raw_bwd(dy, factor, dx):
    """Gradient of the raw bilinear up-sampling.  factor 8: through per-block sums in a workspace (one per device and stream,
    grown to the largest map seen), every element of dy read once."""
    lib = _lib.lib()
    if factor == 8 and UPS8_BLOCK_SUMS:
        key = (dx.t.device, _stream().value)
        need = max(16, lib.xv_upsample_raw_bwd_workspace_bytes(dx.n, dx.h, dx.w, dx.c)) // 4
        ws = _UPS_WS.get(key)
        if ws is None or ws.numel() < need:
            ws = _UPS_WS[key] = torch.empty(need, dtype=torch.float32, device=dx.t.device)
        _lib.check(lib.xv_upsample_raw_bwd_ws(dy.xv(), factor, dx.xv(), _ptr(ws), ws.numel() * 4, _stream()), 'xv_upsample_raw_bwd_ws')
        return dx
    _lib.check(lib.xv_upsample_raw_bwd(dy.xv(), factor, dx.xv(), _stream()), 'xv_upsample_raw_bwd')
    return dx


def bn_apply_ups8(low, st, y, relu=True):
    """y = [relu](bilinear_x8(low) * scale + shift) with the scale / shift bn_forward(ups8_of=low, y=None) left in `st`."""
    _lib.check(_lib.lib().xv_bn_apply_ups8(low.xv(), _ptr(st.scale), _ptr(st.shift), int(bool(relu)), y.xv(), _stream()),
               'xv_bn_apply_ups8')
    return y


def score_dense_fwd_ups8(low, st, w_score, b_score, num_classes, y, score):
    """y = relu(BN(bilinear_x8(low))) with the batch norm's scale / shift in `st` (bn_forward(ups8_of=low, y=None) before this)
    AND score = y . W + b in one launch.  Returns False -- nothing launched -- where the fused kernel does not take the shape."""
    if os.environ.get('XV_FUSED_SCORE_UPS8') == '0':        # A/B timing: the apply pass and the score conv as two launches
        return False
    rc = _lib.lib().xv_score_dense_fwd_ups8(low.xv(), _ptr(st.scale), _ptr(st.shift), _ptr(w_score), _ptr(b_score), num_classes,
                                            y.xv(), _ptr(score), _stream())
    if rc == -2:
        return False
    _lib.check(rc, 'xv_score_dense_fwd_ups8')
    return True


def score_dense_fwd(u, w_score, b_score, num_classes, score):
    _lib.check(_lib.lib().xv_score_dense_fwd(u.xv(), _ptr(w_score), _ptr(b_score), num_classes, _ptr(score), _stream()),
               'xv_score_dense_fwd')
    return score


_CE_WS = {}


def softmax_ce_dense(logits, labels, count, num_classes, loss, dlogits, affine=None):
    """affine (a BnState): `logits` holds raw scores, logits = scores * affine.scale + affine.shift inside the kernel.
    The loss is added up in a fixed order (xv_softmax_ce_dense_ws: per-workgroup partials in a workspace kept per device and
    stream, one small launch adds them): bitwise reproducible, like the rest of the training step."""
    _need(labels, torch.int32, 'labels')
    npix = labels.numel()
    key = (logits.device, _stream().value)
    nbytes = _lib.lib().xv_softmax_ce_dense_workspace_bytes(npix)
    ws = _CE_WS.get(key)
    if ws is None or ws.numel() * 8 < nbytes:
        ws = _CE_WS[key] = torch.empty(max(nbytes // 8, 2048), dtype=torch.float64, device=logits.device)
    _lib.check(_lib.lib().xv_softmax_ce_dense_ws(_ptr(logits), _ptr(affine.scale) if affine is not None else None,
                                                 _ptr(affine.shift) if affine is not None else None, _ptr(labels),
                                                 _ptr(count), num_classes, npix, _ptr(loss), _ptr(dlogits), _ptr(ws),
                                                 ws.numel() * 8, _stream()), 'xv_softmax_ce_dense_ws')
    return dlogits


_SD_WS = {}


def score_dense_bwd(u, dscore, w_score, num_classes, dw_score, db_score, du):
    """Backward of the dense 1x1 score layer at full resolution; filter and bias gradients added in a fixed order (a workspace
    per device and stream, grown to the largest map seen): bitwise reproducible."""
    lib = _lib.lib()
    key = (u.t.device, _stream().value)
    need = max(16, lib.xv_score_dense_bwd_workspace_bytes(u.n, u.h, u.w)) // 4
    ws = _SD_WS.get(key)
    if ws is None or ws.numel() < need:
        ws = _SD_WS[key] = torch.empty(need, dtype=torch.float32, device=u.t.device)
    rc = lib.xv_score_dense_bwd_ws(u.xv(), _ptr(dscore), _ptr(w_score), num_classes, _ptr(dw_score), _ptr(db_score), du.xv(),
                                   _ptr(ws), ws.numel() * 4, _stream())
    _lib.check(rc, 'xv_score_dense_bwd_ws')
    return du
