"""BASELINE.json configs[4] ("fp8 MFMA conv path"): the block-scaled e4m3 convolution
(v_mfma_scale_f32_16x16x128_f8f6f4) through the C ABI against the oracle's 'fp8' policy.

Integer-valued operands on the e4m3 grid make the checks bit-exact: products and fp32 sums of small integers are
exact in any order, so a fragment-map / packing / scale bug cannot hide in a tolerance, and the output conversion
(round-to-nearest-even, saturation at 448, per-tensor power-of-two scale) is compared value for value.
Reference call sites: vgg16.py:7-51, custom_layers.py:124-139 (tf.layers.conv2d), simple_fcn.py:39-79."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import fcn_oracle as fo


@pytest.fixture(scope='module')
def ops():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from modular_semantic_segmentation_amd import ops as _ops
    return _ops


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _oracle(x, wt, b, relu):
    xt = torch.from_numpy(x).permute(0, 3, 1, 2).contiguous()
    return fo.conv2d_same(xt, wt, b, relu=relu)            # NCHW fp32, exact on these operands


def _nhwc(t):
    return t.permute(0, 2, 3, 1).numpy()


CASES = [
    # n, h, w, cin, cout, k, pool, cfg, (ex, ew, ey)
    (1, 16, 16, 128, 64, 3, False, 14, (0, 0, 0)),
    (2, 16, 32, 128, 128, 3, True, 15, (1, -2, 3)),
    (1, 24, 40, 256, 128, 3, True, 14, (-1, 0, 2)),       # partial tiles in both directions
    (1, 6, 10, 512, 256, 3, True, 15, (0, -3, 4)),        # tiny map, four 128-channel chunks
    (2, 32, 64, 128, 64, 3, True, 16, (2, -1, 2)),        # the 8-wave 16x32 tile, five taps per barrier
    (1, 20, 36, 256, 192, 3, False, 16, (0, 0, 1)),       # ... with partial tiles
    (3, 8, 8, 512, 64, 1, False, 14, (0, -2, 0)),         # score_conv shape (1x1, Cout 64)
    (1, 18, 22, 128, 256, 1, False, 15, (1, 1, 1)),
    (1, 16, 32, 384, 128, 3, False, -1, (0, 0, 2)),       # the library's own choice of tile
]


@pytest.mark.parametrize('n,h,w,cin,cout,k,pool,cfg,exps', CASES)
def test_fp8_conv_exact_on_integers(ops, n, h, w, cin, cout, k, pool, cfg, exps):
    ex, ew, ey = exps
    rng = np.random.default_rng(abs(hash((n, h, w, cin, cout, k, cfg))) % 2**32)
    # operands on the e4m3 grid of their scale: integers in [-4, 4] / [-3, 3] times 2^e
    x = (rng.integers(-4, 5, (n, h, w, cin)) * 2.0 ** ex).astype(np.float32)
    wt = (rng.integers(-3, 4, (k, k, cin, cout)) * 2.0 ** ew).astype(np.float32)
    b = (rng.integers(-3, 4, cout) * 2.0 ** (ex + ew)).astype(np.float32)
    xa = ops.Act.from_dense(_dev(x), dtype='fp8', scale_exp=ex)
    assert np.array_equal(xa.real().cpu().numpy(), x)                     # the test input itself is exact
    wp, e_used = ops.pack_conv_weights_f8(_dev(wt), scale_exp=ew)
    assert e_used == ew
    y32 = _oracle(x, wt, b, True)
    # ---- fp8 output (+ fused pool) ---------------------------------------------------------------------------
    y = ops.Act(n, h, w, cout, dtype='fp8', scale_exp=ey)
    q = ops.Act(n, h // 2, w // 2, cout, dtype='fp8', scale_exp=ey) if pool else None
    ops.conv2d_fwd(xa, wp, _dev(b), k, relu=True, y=y, pooled=q, cfg=cfg)
    torch.cuda.synchronize()
    ref = fo.round_e4m3(_nhwc(y32), ey)
    got = y.real().cpu().numpy()
    bad = np.argwhere(got != ref)
    assert bad.size == 0, 'first mismatches (n,y,x,c): %s got %s want %s' % (
        bad[:5].tolist(), got[tuple(bad[:5].T)], ref[tuple(bad[:5].T)])
    raw = y.t.view(torch.uint8).cpu().numpy()
    assert not raw[:, 0].any() and not raw[:, -1].any() and not raw[:, :, 0].any() and not raw[:, :, -1].any()
    if pool:
        refq = fo.round_e4m3(_nhwc(fo.maxpool2(y32)), ey)
        assert np.array_equal(q.real().cpu().numpy(), refq)
        q2 = ops.Act(n, h // 2, w // 2, cout, dtype='fp8', scale_exp=ey)
        ops.conv2d_fwd(xa, wp, _dev(b), k, relu=True, pooled=q2, write_y=False, cfg=cfg)       # pooled-only launch
        torch.cuda.synchronize()
        assert torch.equal(q2.t.view(torch.uint8), q.t.view(torch.uint8))
    # ---- bf16 output from the same fp8 operands (conv4_3 / conv5_3 feed the bf16 decoder) ---------------------
    yb, _ = ops.conv2d_fwd(xa, wp, _dev(b), k, relu=False, cfg=cfg)
    torch.cuda.synchronize()
    refb = fo.round_bf16(_oracle(x, wt, b, False)).permute(0, 2, 3, 1).numpy()
    assert np.array_equal(yb.interior().float().cpu().numpy(), refb)


def test_fp8_conv_tile_configurations_agree(ops):
    """Configurations 14 / 15 / 16 of the fp8 kernel: bit-identical outputs; the others refuse fp8 operands."""
    from modular_semantic_segmentation_amd import _lib
    rng = np.random.default_rng(5)
    n, h, w, cin, cout = 2, 24, 40, 256, 128
    x = rng.integers(-4, 5, (n, h, w, cin)).astype(np.float32)
    wt = rng.integers(-3, 4, (3, 3, cin, cout)).astype(np.float32)
    b = rng.integers(-3, 4, cout).astype(np.float32)
    xa = ops.Act.from_dense(_dev(x), dtype='fp8')
    wp, _ = ops.pack_conv_weights_f8(_dev(wt), scale_exp=0)
    outs = {}
    for cfg in range(_lib.lib().xv_conv2d_num_cfgs()):
        y = ops.Act(n, h, w, cout, dtype='fp8', scale_exp=3)
        try:
            ops.conv2d_fwd(xa, wp, _dev(b), 3, relu=True, y=y, cfg=cfg)
        except _lib.XvError:
            assert cfg not in (14, 15, 16)
            continue
        torch.cuda.synchronize()
        outs[cfg] = y.t.view(torch.uint8).clone()
    assert sorted(outs) == [14, 15, 16]
    assert torch.equal(outs[14], outs[15]) and torch.equal(outs[14], outs[16])


@pytest.mark.parametrize('cfg', [-1, 4, 10, 14, 16])
def test_bf16_conv_with_fp8_output(ops, cfg):
    """conv2_1 of the fp8 network: 64 input channels are half an fp8 MFMA, so it stays a bf16 convolution whose
    epilogue writes e4m3 for conv2_2 (first-generation tiles only; the chooser avoids the others)."""
    rng = np.random.default_rng(9)
    n, h, w, cin, cout, ey = 1, 24, 48, 64, 128, 2
    x = rng.integers(-2, 3, (n, h, w, cin)).astype(np.float32)
    wt = rng.integers(-1, 2, (3, 3, cin, cout)).astype(np.float32)
    b = rng.integers(-3, 4, cout).astype(np.float32)
    xa, wp = ops.Act.from_dense(_dev(x)), ops.pack_conv_weights(_dev(wt))
    y = ops.Act(n, h, w, cout, dtype='fp8', scale_exp=ey)
    q = ops.Act(n, h // 2, w // 2, cout, dtype='fp8', scale_exp=ey)
    ops.conv2d_fwd(xa, wp, _dev(b), 3, relu=True, y=y, pooled=q, cfg=cfg)
    torch.cuda.synchronize()
    y32 = _oracle(x, wt, b, True)
    assert np.array_equal(y.real().cpu().numpy(), fo.round_e4m3(_nhwc(y32), ey))
    assert np.array_equal(q.real().cpu().numpy(), fo.round_e4m3(_nhwc(fo.maxpool2(y32)), ey))
    from modular_semantic_segmentation_amd import _lib
    with pytest.raises(_lib.XvError):
        ops.conv2d_fwd(xa, wp, _dev(b), 3, relu=True, y=y, cfg=17)        # the all-DMA kernel writes bf16 only


def test_fp8_conv_random_operands_and_weight_packing(ops):
    """Random fp32 weights / activations: the packer's quantisation (round-to-nearest-even after the power-of-two
    scale, chosen from max|w|) equals the oracle's, so the conv of the quantised operands agrees to fp32
    accumulation-order noise before the output rounding; the e4m3 output then differs by at most one grid step on the
    few values that sit on a rounding boundary."""
    rng = np.random.default_rng(11)
    n, h, w, cin, cout = 2, 32, 48, 256, 128
    x = np.maximum(rng.standard_normal((n, h, w, cin)), 0).astype(np.float32) * 37.0
    wt = (rng.standard_normal((3, 3, cin, cout)) * 0.02).astype(np.float32)
    b = (rng.standard_normal(cout) * 0.5).astype(np.float32)
    ex = fo.fp8_scale_exp(np.abs(x).max())
    xq = fo.round_e4m3(x, ex)
    xa = ops.Act.from_dense(_dev(x), dtype='fp8', scale_exp=ex)
    assert np.array_equal(xa.real().cpu().numpy(), xq)
    wp, ew = ops.pack_conv_weights_f8(_dev(wt))
    assert ew == fo.fp8_scale_exp(np.abs(wt).max())
    wq = fo.round_e4m3(wt, ew)
    y32 = _nhwc(_oracle(xq, wq, b, True))
    yb, _ = ops.conv2d_fwd(xa, wp, _dev(b), 3, relu=True)
    torch.cuda.synchronize()
    got = yb.interior().float().cpu().numpy()
    np.testing.assert_allclose(got, y32, rtol=2.0 ** -8, atol=1e-4 * np.abs(y32).max())
    ey = fo.fp8_scale_exp(np.abs(y32).max(), 1)
    y8 = ops.Act(n, h, w, cout, dtype='fp8', scale_exp=ey)
    ops.conv2d_fwd(xa, wp, _dev(b), 3, relu=True, y=y8)
    torch.cuda.synchronize()
    got8, ref8 = y8.real().cpu().numpy(), fo.round_e4m3(y32, ey)
    assert (got8 != ref8).mean() < 2e-3
    np.testing.assert_allclose(got8, ref8, rtol=2.0 ** -3, atol=2.0 ** (ey - 9))
