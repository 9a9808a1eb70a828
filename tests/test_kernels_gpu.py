"""Parity of every HIP kernel (called through the C ABI) against the CPU oracle.
Integer-valued operands make the conv checks bit-exact (fp32 sums of small integers are exact
in any order), so a layout / swizzle / fragment-map bug cannot hide inside a tolerance."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import fcn_oracle as fo
from oracle import fusion_oracle as fu


@pytest.fixture(scope='module')
def ops():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from modular_semantic_segmentation_amd import ops as _ops
    return _ops


def _dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def _conv_oracle(x, w, b, relu, k):
    xt = torch.from_numpy(x).permute(0, 3, 1, 2).contiguous()
    y = fo.conv2d_same(xt, w, b, relu=relu)
    return y, fo.round_bf16(y).permute(0, 2, 3, 1).numpy()


CONV_CASES = [
    # n, h, w, cin, cout, k, pool
    (1, 16, 16, 64, 64, 3, False),
    (2, 16, 32, 64, 128, 3, True),
    (1, 24, 40, 128, 128, 3, False),    # partial tiles in both directions
    (1, 6, 10, 256, 512, 3, True),      # tiny image, many channel chunks
    (3, 8, 8, 512, 64, 1, False),       # score_conv shape (1x1, Cout 64)
    (1, 18, 22, 128, 256, 1, False),
    (1, 32, 48, 64, 64, 3, True),       # conv1_2-like with fused pool1
]


@pytest.mark.parametrize('n,h,w,cin,cout,k,pool', CONV_CASES)
def test_conv2d_mfma_exact_on_integers(ops, n, h, w, cin, cout, k, pool):
    rng = np.random.default_rng(hash((n, h, w, cin, cout, k)) % 2**32)
    x = rng.integers(-2, 3, (n, h, w, cin)).astype(np.float32)
    wt = rng.integers(-1, 2, (k, k, cin, cout)).astype(np.float32)
    b = rng.integers(-3, 4, cout).astype(np.float32)
    xa = ops.Act.from_dense(_dev(x))
    wp = ops.pack_conv_weights(_dev(wt))
    q = ops.Act(n, h // 2, w // 2, cout) if pool else None
    y, _ = ops.conv2d_fwd(xa, wp, _dev(b), k, relu=True, pooled=q)
    torch.cuda.synchronize()
    y32, ref = _conv_oracle(x, wt, b, True, k)
    got = y.interior().float().cpu().numpy()
    bad = np.argwhere(got != ref)
    assert bad.size == 0, 'first mismatches (n,y,x,c): %s got %s want %s' % (
        bad[:5].tolist(), got[tuple(bad[:5].T)], ref[tuple(bad[:5].T)])
    # the zero border must be untouched
    full = y.t.float().cpu().numpy()
    assert np.all(full[:, 0] == 0) and np.all(full[:, -1] == 0) and np.all(full[:, :, 0] == 0) and np.all(full[:, :, -1] == 0)
    if pool:
        refq = fo.round_bf16(fo.maxpool2(y32)).permute(0, 2, 3, 1).numpy()
        gotq = q.interior().float().cpu().numpy()
        assert np.array_equal(gotq, refq)
        # standalone pool kernel agrees with the fused epilogue
        q2 = ops.maxpool2x2_fwd(y)
        torch.cuda.synchronize()
        assert torch.equal(q2.t, q.t)
        # pooled-only launch (no full-resolution store)
        q3 = ops.Act(n, h // 2, w // 2, cout)
        ops.conv2d_fwd(xa, wp, _dev(b), k, relu=True, pooled=q3, write_y=False)
        torch.cuda.synchronize()
        assert torch.equal(q3.t, q.t)


@pytest.mark.parametrize('k', [3, 1])
def test_conv2d_every_tile_configuration(ops, k):
    """All tile configurations compute bit-identical results (same accumulation order)."""
    from modular_semantic_segmentation_amd import _lib
    rng = np.random.default_rng(k)
    n, h, w, cin, cout = 2, 24, 40, 128, 256
    x = rng.integers(-2, 3, (n, h, w, cin)).astype(np.float32)
    wt = rng.integers(-1, 2, (k, k, cin, cout)).astype(np.float32)
    b = rng.integers(-3, 4, cout).astype(np.float32)
    xa, wp, bd = ops.Act.from_dense(_dev(x)), ops.pack_conv_weights(_dev(wt)), _dev(b)
    y32, ref = _conv_oracle(x, wt, b, True, k)
    refq = fo.round_bf16(fo.maxpool2(y32)).permute(0, 2, 3, 1).numpy()
    ran = 0
    for cfg in range(_lib.lib().xv_conv2d_num_cfgs()):
        q = ops.Act(n, h // 2, w // 2, cout) if k == 3 and cfg != 22 else None   # 22: three rows per wave, no fused pool
        try:
            y, _ = ops.conv2d_fwd(xa, wp, bd, k, relu=True, pooled=q, cfg=cfg)
        except _lib.XvError:
            # generation 2 / 2b: 3x3 only, generation 3: 1x1 only, its narrow form: 1x1 onto 64 channels only
            # generation 4: 3x3 only; 24 takes e4m3 maps only
            # generation 5 (27 / 28): 3x3 on maps that tile exactly in 24x16 / 32x16 only
            retired = cfg in (0, 19, 20, 21, 25) or 2 <= cfg <= 13     # round 4: 25; round 5: every variant nothing selects
            assert retired or (cfg in (17, 22, 26) and k == 1) or (cfg == 18 and k == 3) or cfg in (23, 24, 27, 28)
            continue
        ran += 1
        torch.cuda.synchronize()
        assert np.array_equal(y.interior().float().cpu().numpy(), ref), 'cfg %d' % cfg
        if q is not None:
            assert np.array_equal(q.interior().float().cpu().numpy(), refq), 'cfg %d pooled' % cfg
    assert ran == (7 if k == 3 else 5)      # 3x3: 1, 14-17, 22, 26; 1x1: 1, 14-16, 18


@pytest.mark.parametrize('n,h,w,cin', [(1, 24, 48, 512), (16, 24, 48, 512), (1, 48, 96, 512), (3, 7, 5, 128), (2, 30, 33, 192)])
def test_conv1x1_narrow_gemm(ops, n, h, w, cin):
    """Tile configuration 23 (the FCN's score convs, simple_fcn.py:69-79: 1x1 onto 64 channels): exact on integers, border
    untouched, row counts that are not multiples of the 64-row tile, the library's own choice for this shape, and the
    same bits as a first-generation tile."""
    rng = np.random.default_rng(n * h * w + cin)
    x = rng.integers(-2, 3, (n, h, w, cin)).astype(np.float32)
    wt = rng.integers(-1, 2, (1, 1, cin, 64)).astype(np.float32)
    b = rng.integers(-3, 4, 64).astype(np.float32)
    xa, wp, bd = ops.Act.from_dense(_dev(x)), ops.pack_conv_weights(_dev(wt)), _dev(b)
    _, ref = _conv_oracle(x, wt, b, True, 1)
    for cfg in (23, -1):
        y, _ = ops.conv2d_fwd(xa, wp, bd, 1, relu=True, cfg=cfg)
        torch.cuda.synchronize()
        assert np.array_equal(y.interior().float().cpu().numpy(), ref), cfg
        for edge in (y.t[:, 0], y.t[:, -1], y.t[:, :, 0], y.t[:, :, -1]):
            assert not edge.any()
    g = torch.Generator(device='cuda').manual_seed(1)
    xr = ops.Act.from_dense(torch.randn((n, h, w, cin), device='cuda', generator=g))
    wr = ops.pack_conv_weights(torch.randn((1, 1, cin, 64), device='cuda', generator=g) * cin ** -0.5)
    y0, _ = ops.conv2d_fwd(xr, wr, bd, 1, relu=False, cfg=23)
    y1, _ = ops.conv2d_fwd(xr, wr, bd, 1, relu=False, cfg=14)
    torch.cuda.synchronize()
    assert torch.equal(y0.t, y1.t)


@pytest.mark.parametrize('n,h,w,cin,cout', [(1, 6, 10, 64, 128), (2, 24, 48, 576, 256), (3, 7, 5, 128, 384),
                                            (1, 30, 33, 256, 128), (4, 95, 190, 192, 128), (5, 47, 97, 64, 256),
                                            (3, 96, 191, 128, 384)])
def test_conv1x1_flat_gemm(ops, n, h, w, cin, cout):
    """Generation 3 (flat GEMM over the padded rows, cfg 18): exact on integers, with activation + addend + mask in
    the order of the other kernels, border untouched, row counts that are not multiples of the 128-row tile.  The last three
    shapes make enough workgroups for the WIDE form of round 6 (256 rows x 128 channels, three stages, conv1x1_gemm_wide_kernel:
    one, three and six K steps; row counts that are not multiples of 256) -- the same bits as generation 1."""
    rng = np.random.default_rng(n * h * w + cin)
    x = rng.integers(-2, 3, (n, h, w, cin)).astype(np.float32)
    wt = rng.integers(-1, 2, (1, 1, cin, cout)).astype(np.float32)
    b = rng.integers(-3, 4, cout).astype(np.float32)
    xa, wp, bd = ops.Act.from_dense(_dev(x)), ops.pack_conv_weights(_dev(wt)), _dev(b)
    y, _ = ops.conv2d_fwd(xa, wp, bd, 1, relu=True, cfg=18)
    torch.cuda.synchronize()
    _, ref = _conv_oracle(x, wt, b, True, 1)
    assert np.array_equal(y.interior().float().cpu().numpy(), ref)
    for edge in (y.t[:, 0], y.t[:, -1], y.t[:, :, 0], y.t[:, :, -1]):
        assert not edge.any()
    y0, _ = ops.conv2d_fwd(xa, wp, bd, 1, relu=False, cfg=18)
    y1, _ = ops.conv2d_fwd(xa, wp, bd, 1, relu=False, cfg=14)
    torch.cuda.synchronize()
    assert torch.equal(y0.t, y1.t)


@pytest.mark.parametrize('gen2', [17])
@pytest.mark.parametrize('shape', [(2, 16, 32, 64, 64), (3, 32, 64, 128, 192), (1, 48, 96, 32, 64),
                                   (2, 24, 40, 64, 64), (1, 20, 36, 128, 64), (2, 6, 10, 64, 128),
                                   (9, 64, 96, 256, 128)])
def test_conv2d_generation2_all_dma(ops, shape, gen2):
    """The all-LDS-DMA kernel of generation 2 (32-channel chunks, its own packed image; 17 = conv_dma_kernel -- 21, the variant
    with the item barrier inside the last tap, was retired in round 5: nothing selected it; the last shape gives every
    workgroup several tiles of several chunks) against the oracle, bit for bit on
    integer operands: full output, fused pool, pooled-only launch, untouched border -- whole 16x32 tiles and
    partial ones (clamped DMA offsets, predicated stores)."""
    from modular_semantic_segmentation_amd import _lib
    n, h, w, cin, cout = shape
    rng = np.random.default_rng(sum(shape))
    x = rng.integers(-2, 3, (n, h, w, cin)).astype(np.float32)
    wt = rng.integers(-1, 2, (3, 3, cin, cout)).astype(np.float32)
    b = rng.integers(-3, 4, cout).astype(np.float32)
    if cin % 64:
        pytest.skip('the packed buffer carries both images; 64-channel multiples only')
    xa, wp, bd = ops.Act.from_dense(_dev(x)), ops.pack_conv_weights(_dev(wt)), _dev(b)
    y32, ref = _conv_oracle(x, wt, b, True, 3)
    refq = fo.round_bf16(fo.maxpool2(y32)).permute(0, 2, 3, 1).numpy()
    q = ops.Act(n, h // 2, w // 2, cout)
    y, _ = ops.conv2d_fwd(xa, wp, bd, 3, relu=True, pooled=q, cfg=gen2)
    torch.cuda.synchronize()
    assert np.array_equal(y.interior().float().cpu().numpy(), ref)
    assert np.array_equal(q.interior().float().cpu().numpy(), refq)
    full = y.t.float().cpu().numpy()
    assert np.all(full[:, 0] == 0) and np.all(full[:, -1] == 0) and np.all(full[:, :, 0] == 0) and np.all(full[:, :, -1] == 0)
    q2 = ops.Act(n, h // 2, w // 2, cout)
    ops.conv2d_fwd(xa, wp, bd, 3, relu=True, pooled=q2, write_y=False, cfg=gen2)
    torch.cuda.synchronize()
    assert torch.equal(q2.t, q.t)


@pytest.mark.parametrize('shape', [(1, 16, 32, 64, 64), (2, 32, 64, 128, 192), (1, 48, 96, 256, 64), (5, 96, 192, 64, 128),
                                   (2, 24, 40, 64, 64), (1, 20, 36, 128, 64), (2, 6, 10, 64, 128), (3, 24, 48, 512, 128)])
@pytest.mark.parametrize('cfg', [26])
def test_conv2d_generation4_bf16(ops, shape, cfg):
    """Configuration 26 (conv_dma4_kernel<false, .., M16>, conv_f8_dma.hip: 16x16x32 bf16 MFMA blocks, all operands
    by LDS-DMA, its own packed image; the last shape gives every workgroup several two-chunk tiles) against the
    oracle, bit for bit on
    integer operands: full output, fused pool, pooled-only launch, no relu, untouched border -- whole 16x32 tiles and
    partial ones (clamped DMA offsets, predicated stores)."""
    from modular_semantic_segmentation_amd import _lib
    n, h, w, cin, cout = shape
    rng = np.random.default_rng(sum(shape))
    x = rng.integers(-2, 3, (n, h, w, cin)).astype(np.float32)
    wt = rng.integers(-1, 2, (3, 3, cin, cout)).astype(np.float32)
    b = rng.integers(-3, 4, cout).astype(np.float32)
    xa, wp, bd = ops.Act.from_dense(_dev(x)), ops.pack_conv_weights(_dev(wt)), _dev(b)
    y32, ref = _conv_oracle(x, wt, b, True, 3)
    refq = fo.round_bf16(fo.maxpool2(y32)).permute(0, 2, 3, 1).numpy()
    q = ops.Act(n, h // 2, w // 2, cout)
    y, _ = ops.conv2d_fwd(xa, wp, bd, 3, relu=True, pooled=q, cfg=cfg)
    torch.cuda.synchronize()
    assert np.array_equal(y.interior().float().cpu().numpy(), ref)
    assert np.array_equal(q.interior().float().cpu().numpy(), refq)
    full = y.t.float().cpu().numpy()
    assert np.all(full[:, 0] == 0) and np.all(full[:, -1] == 0) and np.all(full[:, :, 0] == 0) and np.all(full[:, :, -1] == 0)
    q2 = ops.Act(n, h // 2, w // 2, cout)
    ops.conv2d_fwd(xa, wp, bd, 3, relu=True, pooled=q2, write_y=False, cfg=cfg)
    torch.cuda.synchronize()
    assert torch.equal(q2.t, q.t)
    _, refn = _conv_oracle(x, wt, b, False, 3)
    yn, _ = ops.conv2d_fwd(xa, wp, bd, 3, relu=False, cfg=cfg)
    torch.cuda.synchronize()
    assert np.array_equal(yn.interior().float().cpu().numpy(), refn)


@pytest.mark.parametrize('shape,cfg', [((16, 24, 48, 512, 512), 27), ((1, 24, 16, 64, 64), 27), ((3, 48, 32, 128, 192), 27),
                                       ((40, 24, 48, 64, 128), 27), ((2, 72, 16, 192, 64), 27),
                                       ((1, 32, 16, 64, 64), 28), ((2, 64, 48, 128, 128), 28), ((5, 96, 192, 64, 128), 28),
                                       ((3, 32, 64, 256, 192), 28), ((30, 32, 32, 64, 64), 28)])
def test_conv2d_generation5_column_tiles(ops, shape, cfg):
    """Configurations 27 / 28 (conv_dma5_kernel<3 / 4>, conv_col_dma.hip: the generation-4 loop on a column of 8 waves x 3 / 4
    rows x 16 columns; 24x16 tiles cover the 24x48 conv5 maps exactly) against the oracle, bit for bit on integer operands:
    full output, no relu, untouched border, one to many tiles per workgroup, one- and two-chunk layers with resident
    weights; 28 also the fused pool and the pooled-only launch; 27 refuses a pooled output, both refuse maps that do not tile."""
    from modular_semantic_segmentation_amd import _lib
    n, h, w, cin, cout = shape
    rng = np.random.default_rng(sum(shape) + cfg)
    x = rng.integers(-2, 3, (n, h, w, cin)).astype(np.float32)
    wt = rng.integers(-1, 2, (3, 3, cin, cout)).astype(np.float32)
    b = rng.integers(-3, 4, cout).astype(np.float32)
    xa, wp, bd = ops.Act.from_dense(_dev(x)), ops.pack_conv_weights(_dev(wt)), _dev(b)
    y32, ref = _conv_oracle(x, wt, b, True, 3)
    y, _ = ops.conv2d_fwd(xa, wp, bd, 3, relu=True, cfg=cfg)
    torch.cuda.synchronize()
    assert np.array_equal(y.interior().float().cpu().numpy(), ref)
    full = y.t.float().cpu().numpy()
    assert np.all(full[:, 0] == 0) and np.all(full[:, -1] == 0) and np.all(full[:, :, 0] == 0) and np.all(full[:, :, -1] == 0)
    _, refn = _conv_oracle(x, wt, b, False, 3)
    yn, _ = ops.conv2d_fwd(xa, wp, bd, 3, relu=False, cfg=cfg)
    torch.cuda.synchronize()
    assert np.array_equal(yn.interior().float().cpu().numpy(), refn)
    if cfg == 28:
        refq = fo.round_bf16(fo.maxpool2(y32)).permute(0, 2, 3, 1).numpy()
        q = ops.Act(n, h // 2, w // 2, cout)
        y2, _ = ops.conv2d_fwd(xa, wp, bd, 3, relu=True, pooled=q, cfg=cfg)
        q2 = ops.Act(n, h // 2, w // 2, cout)
        ops.conv2d_fwd(xa, wp, bd, 3, relu=True, pooled=q2, write_y=False, cfg=cfg)
        torch.cuda.synchronize()
        assert torch.equal(y2.t, y.t)
        assert np.array_equal(q.interior().float().cpu().numpy(), refq) and torch.equal(q2.t, q.t)
    else:
        with pytest.raises(_lib.XvError):
            ops.conv2d_fwd(xa, wp, bd, 3, relu=True, pooled=ops.Act(n, h // 2, w // 2, cout), cfg=cfg)
    xs = ops.Act.from_dense(_dev(x[:, :h - 8]))
    with pytest.raises(_lib.XvError):
        ops.conv2d_fwd(xs, wp, bd, 3, relu=True, cfg=cfg)          # 8 rows short of a tiling


@pytest.mark.parametrize('shape,mode', [((3, 32, 64, 128, 128), 'both'), ((16, 48, 96, 256, 512), 'y'), ((2, 16, 32, 64, 64), 'pool'),
                                        ((5, 32, 32, 192, 64), 'both'), ((16, 24, 48, 512, 512), 'y'), ((3, 48, 16, 128, 128), 'y'),
                                        ((1, 16, 32, 128, 64), 'y'), ((37, 16, 32, 64, 128), 'pool')])
def test_conv2d_pair_equals_two_launches(ops, shape, mode):
    """xv_conv2d_fwd_pair (the same layer of the two experts in ONE launch of the generation-4 / 5 kernel: concatenated tile
    lists, per-tile maps / weights / bias) against two xv_conv2d_fwd launches, bit for bit on random real-valued operands --
    full maps, fused pools, pooled-only; fewer tiles than workgroups up to many rounds; the border of every map untouched."""
    n, h, w, cin, cout = shape
    g = torch.Generator(device='cuda').manual_seed(sum(shape))
    outs = {}
    X, Wp, B = [], [], []
    for e in range(2):
        X.append(ops.Act.from_dense(torch.randn(n, h, w, cin, device='cuda', generator=g)))
        Wp.append(ops.pack_conv_weights(torch.randn(3, 3, cin, cout, device='cuda', generator=g) * (9 * cin) ** -0.5))
        B.append(torch.randn(cout, device='cuda', generator=g))

    def outputs():
        y = [ops.Act(n, h, w, cout) for _ in range(2)] if mode != 'pool' else [None, None]
        q = [ops.Act(n, h // 2, w // 2, cout) for _ in range(2)] if mode != 'y' else [None, None]
        return y, q
    y1, q1 = outputs()
    for e in range(2):
        ops.conv2d_fwd(X[e], Wp[e], B[e], 3, relu=True, y=y1[e], pooled=q1[e], write_y=y1[e] is not None)
    y2, q2 = outputs()
    assert ops.conv2d_fwd_pair(X[0], Wp[0], B[0], X[1], Wp[1], B[1], relu=True, ya=y2[0], yb=y2[1], pa=q2[0], pb=q2[1])
    torch.cuda.synchronize()
    for e in range(2):
        for a, b in ((y1[e], y2[e]), (q1[e], q2[e])):
            if a is not None:
                assert torch.equal(a.t, b.t)
                assert float(a.interior().float().abs().sum()) > 0
    if mode == 'y':
        assert not torch.equal(y2[0].t, y2[1].t)


def test_conv2d_pair_refuses_what_the_kernels_do_not_take(ops):
    """Maps that do not tile (generation 2 would run them), a pooled output on the 24x16 tile, mismatched twins: False,
    nothing launched."""
    def mk(n, h, w, cin, cout):
        return (ops.Act.from_dense(torch.randn(n, h, w, cin, device='cuda')),
                ops.pack_conv_weights(torch.randn(3, 3, cin, cout, device='cuda') * 0.05), torch.zeros(cout, device='cuda'))
    xa, wa, ba = mk(2, 20, 24, 64, 64)
    y = [ops.Act(2, 20, 24, 64) for _ in range(2)]
    assert not ops.conv2d_fwd_pair(xa, wa, ba, xa, wa, ba, ya=y[0], yb=y[1])
    assert float(y[0].t.float().abs().sum()) == 0
    xa, wa, ba = mk(2, 24, 48, 64, 64)
    q = [ops.Act(2, 12, 24, 64) for _ in range(2)]
    assert not ops.conv2d_fwd_pair(xa, wa, ba, xa, wa, ba, pa=q[0], pb=q[1])
    xb, wb, bb = mk(2, 32, 64, 64, 64)
    xc, wc, bc = mk(3, 32, 64, 64, 64)
    assert not ops.conv2d_fwd_pair(xb, wb, bb, xc, wc, bc, ya=ops.Act(2, 32, 64, 64), yb=ops.Act(3, 32, 64, 64))


@pytest.mark.parametrize('shape', [(2, 24, 48, 512, 128), (1, 24, 16, 64, 64), (3, 30, 40, 128, 64), (2, 48, 20, 64, 192),
                                   (40, 24, 48, 128, 128), (1, 6, 10, 64, 64)])
def test_conv2d_generation2_three_row_tile(ops, shape):
    """Configuration 22 (conv_dma_kernel on a 24x16 tile, three rows per wave -- the 24x48 conv5 maps): bit for bit against
    the oracle on integer operands, whole and partial tiles, several tiles per workgroup, border untouched; a pooled
    output is refused; the data-gradient epilogue (addend + relu mask) equals generation 1's."""
    from modular_semantic_segmentation_amd import _lib
    n, h, w, cin, cout = shape
    rng = np.random.default_rng(sum(shape))
    x = rng.integers(-2, 3, (n, h, w, cin)).astype(np.float32)
    wt = rng.integers(-1, 2, (3, 3, cin, cout)).astype(np.float32)
    b = rng.integers(-3, 4, cout).astype(np.float32)
    xa, wp, bd = ops.Act.from_dense(_dev(x)), ops.pack_conv_weights(_dev(wt)), _dev(b)
    for relu in (True, False):
        _, ref = _conv_oracle(x, wt, b, relu, 3)
        y, _ = ops.conv2d_fwd(xa, wp, bd, 3, relu=relu, cfg=22)
        torch.cuda.synchronize()
        assert np.array_equal(y.interior().float().cpu().numpy(), ref)
        full = y.t.float().cpu().numpy()
        assert np.all(full[:, 0] == 0) and np.all(full[:, -1] == 0) and np.all(full[:, :, 0] == 0) and np.all(full[:, :, -1] == 0)
    if h % 2 == 0 and w % 2 == 0:
        with pytest.raises(_lib.XvError):
            ops.conv2d_fwd(xa, wp, bd, 3, relu=True, pooled=ops.Act(n, h // 2, w // 2, cout), cfg=22)


@pytest.mark.parametrize('shape', [(2, 32, 64, 128, 64), (3, 16, 32, 64, 192), (1, 48, 96, 256, 128), (4, 24, 48, 128, 128),
                                   (2, 48, 16, 64, 64), (5, 24, 16, 192, 64), (1, 20, 36, 64, 64)])
def test_data_gradient_epilogue_on_generation4_and_5(ops, shape):
    """xv_conv2d_bwd_data (Conv2DBackpropInput + AddN + ReluGrad, base_model.py:153-162 over simple_fcn.py:39-79) on maps that
    tile exactly: the library now takes the 16x16 forms of generation 4 (configuration 26, 16x32 tiles) and generation 5 (27,
    24x16 tiles: the conv5 maps) with the addend + relu-mask epilogue on packed bf16 pairs.  Against the plain convolution of
    a first-generation tile + the same arithmetic in float32 (one rounding), bit for bit on integer operands; every
    combination of addend / mask; the chooser's pick is asserted so that the test cannot pass on a fallback."""
    from modular_semantic_segmentation_amd import _lib
    n, h, w, cin, cout = shape            # of the FORWARD conv: dy has cout channels, dx has cin
    rng = np.random.default_rng(sum(shape))
    wt = torch.from_numpy(rng.integers(-1, 2, (3, 3, cin, cout)).astype(np.float32)).cuda()
    wd = ops.pack_conv_weights_dgrad(wt)
    dy = ops.Act.from_dense(_dev(rng.integers(-2, 3, (n, h, w, cout)).astype(np.float32)))
    ref = ops.Act.from_dense(_dev(rng.integers(-1, 3, (n, h, w, cin)).astype(np.float32) * 0.5))
    ref.interior()[0, 0, 0, :8] = torch.tensor([0.0, -0.0, 1e-30, -1e-30, 3.0, -3.0, 0.0078125, -0.0078125], device='cuda').to(torch.bfloat16)
    add = ops.Act.from_dense(_dev(rng.integers(-3, 4, (n, h, w, cin)).astype(np.float32)))
    zb = torch.zeros(cin, device='cuda')
    want_cfg = 26 if (h % 16 == 0 and w % 32 == 0) else (27 if (h % 24 == 0 and w % 16 == 0) else None)
    if want_cfg == 26 and h % 24 == 0 and w % 16 == 0:
        # the latency rule (round 5): below two rounds of 16x32 items the 24x16 tile's smaller items end sooner
        cus = torch.cuda.get_device_properties(0).multi_processor_count
        i16, i24 = n * (h // 16) * (w // 32) * (cin // 64), n * (h // 24) * (w // 16) * (cin // 64)
        if i16 < 2 * cus and -(-i24 // cus) * 384 / 0.95 < -(-i16 // cus) * 512:
            want_cfg = 27
    got_cfg = _lib.lib().xv_conv2d_choose_cfg(n, h, w, cout, cin, 3, 0, 0, 2)
    if want_cfg is not None:
        assert got_cfg == want_cfg, (got_cfg, want_cfg)
    else:
        assert got_cfg not in (26, 27)
    plain, _ = ops.conv2d_fwd(dy, wd, zb, 3, relu=False, cfg=14)
    torch.cuda.synchronize()
    base = plain.interior().float()
    for use_add, use_ref in ((True, True), (False, True), (True, False), (False, False)):
        dx = ops.conv2d_bwd_data(dy, wd, zb, ops.Act(n, h, w, cin), 3, relu_ref=ref if use_ref else None,
                                 addend=add if use_add else None)
        torch.cuda.synchronize()
        v = base + (add.interior().float() if use_add else 0.0)
        if use_ref:
            v = v * (ref.interior().float() > 0)
        assert torch.equal(dx.interior().float(), v.to(torch.bfloat16).float()), (use_add, use_ref)
        full = dx.t.float()
        assert float(full[:, 0].abs().max()) == 0 and float(full[:, -1].abs().max()) == 0
        assert float(full[:, :, 0].abs().max()) == 0 and float(full[:, :, -1].abs().max()) == 0


@pytest.mark.parametrize('n,h,w,cin', [(1, 16, 32, 3), (2, 32, 64, 1), (3, 48, 96, 3), (1, 64, 32, 1), (5, 96, 192, 3), (16, 32, 64, 1)])
def test_first_pair_fused_equals_the_two_kernels(ops, n, h, w, cin):
    """xv_conv_first_pair_fwd (conv1_1 evaluated into conv1_2's LDS patch buffers, simple_fcn.py:39-41) against
    xv_conv2d_first_fwd + xv_conv2d_fwd: the same bits -- full map, fused pool, pooled-only launch, with and without relu --
    on raw-image-like inputs (0 .. 255 / 0 .. 65535 floats, image borders inside and at the edge of tiles), and against the
    float64 oracle of both layers within the bf16 storage tolerance; maps that do not tile are refused."""
    rng = np.random.default_rng(n * h + w + cin)
    x = rng.integers(0, 256 if cin == 3 else 65536, (n, h, w, cin)).astype(np.float32)
    w1 = (rng.standard_normal((3, 3, cin, 64)) * (0.02 if cin == 3 else 0.02 / 256)).astype(np.float32)
    b1 = (rng.standard_normal(64) * 0.1).astype(np.float32)
    w2 = (rng.standard_normal((3, 3, 64, 64)) / np.sqrt(9 * 64)).astype(np.float32)
    b2 = (rng.standard_normal(64) * 0.1).astype(np.float32)
    xd, w1d, b1d, b2d = _dev(x), _dev(w1), _dev(b1), _dev(b2)
    w2p = ops.pack_conv_weights(_dev(w2))
    for relu2 in (True, False):
        y1 = ops.conv2d_first_fwd(xd, w1d, b1d, ops.Act(n, h, w, 64))
        q_ref = ops.Act(n, h // 2, w // 2, 64)
        y_ref, _ = ops.conv2d_fwd(y1, w2p, b2d, 3, relu=relu2, pooled=q_ref)
        y, q, q2 = ops.Act(n, h, w, 64), ops.Act(n, h // 2, w // 2, 64), ops.Act(n, h // 2, w // 2, 64)
        assert ops.conv_first_pair_fwd(xd, w1d, b1d, w2p, b2d, y=y, pooled=q, relu2=relu2)
        assert ops.conv_first_pair_fwd(xd, w1d, b1d, w2p, b2d, pooled=q2, relu2=relu2)
        torch.cuda.synchronize()
        assert torch.equal(y.t, y_ref.t), 'full map differs (relu2=%s): %d values' % (relu2, int((y.t != y_ref.t).sum()))
        assert torch.equal(q.t, q_ref.t) and torch.equal(q2.t, q_ref.t)
    # the e4m3-out form (the first e4m3 map of the fp8 graph): the same bytes as the two kernels onto e4m3 maps, with the
    # scale that keeps the map inside +-448 and one that saturates
    for scale_exp in (-3, -7):
        y1 = ops.conv2d_first_fwd(xd, w1d, b1d, ops.Act(n, h, w, 64))
        q_ref = ops.Act(n, h // 2, w // 2, 64, dtype='fp8', scale_exp=scale_exp)
        y_ref = ops.Act(n, h, w, 64, dtype='fp8', scale_exp=scale_exp)
        ops.conv2d_fwd(y1, w2p, b2d, 3, relu=True, y=y_ref, pooled=q_ref)
        y8, q8 = ops.Act(n, h, w, 64, dtype='fp8', scale_exp=scale_exp), ops.Act(n, h // 2, w // 2, 64, dtype='fp8', scale_exp=scale_exp)
        q8b = ops.Act(n, h // 2, w // 2, 64, dtype='fp8', scale_exp=scale_exp)
        assert ops.conv_first_pair_fwd(xd, w1d, b1d, w2p, b2d, y=y8, pooled=q8)
        assert ops.conv_first_pair_fwd(xd, w1d, b1d, w2p, b2d, pooled=q8b)
        torch.cuda.synchronize()
        assert torch.equal(y8.t.view(torch.uint8), y_ref.t.view(torch.uint8))
        assert torch.equal(q8.t.view(torch.uint8), q_ref.t.view(torch.uint8)) and torch.equal(q8b.t.view(torch.uint8), q_ref.t.view(torch.uint8))
        assert int((q8.t.view(torch.uint8) != 0).sum()) > 0
    # oracle: conv1_1 in float32 -> bf16, conv1_2 with bf16 operands and fp32 accumulation -> bf16
    xo = torch.from_numpy(x).permute(0, 3, 1, 2).contiguous()
    a1 = fo.round_bf16(fo.conv2d_same(xo, w1, b1, relu=True))
    a2 = fo.round_bf16(fo.conv2d_same(a1, fo.round_bf16(torch.from_numpy(w2)).numpy(), b2, relu=False)).permute(0, 2, 3, 1).numpy()
    got = y.interior().float().cpu().numpy()
    assert np.abs(got - a2).max() <= 2e-2 * np.abs(a2).max()
    assert not ops.conv_first_pair_fwd(_dev(x[:, :h - 8]), w1d, b1d, w2p, b2d, y=ops.Act(n, h - 8, w, 64))


def test_conv2d_mfma_random_bf16(ops):
    """Random bf16 operands: fp32-accumulate result within accumulation-order tolerance."""
    rng = np.random.default_rng(7)
    n, h, w, cin, cout = 2, 24, 32, 256, 256
    x = fo.round_bf16(rng.standard_normal((n, h, w, cin)).astype(np.float32))
    wt = fo.round_bf16((rng.standard_normal((3, 3, cin, cout)) / np.sqrt(9 * cin)).astype(np.float32))
    b = rng.standard_normal(cout).astype(np.float32)
    y, _ = ops.conv2d_fwd(ops.Act.from_dense(_dev(x)), ops.pack_conv_weights(_dev(wt)), _dev(b), 3, relu=False)
    torch.cuda.synchronize()
    y32, _ = _conv_oracle(x, wt, b, False, 3)
    ref = y32.permute(0, 2, 3, 1).numpy()
    got = y.interior().float().cpu().numpy()
    # bf16 output rounding (2^-9 relative) dominates; fp32 accumulation order adds ~1e-6
    np.testing.assert_allclose(got, ref, rtol=2 ** -8, atol=2e-3)


@pytest.mark.parametrize('relu', [True, False])
@pytest.mark.parametrize('shape', [(2, 20, 28), (3, 23, 50), (1, 64, 272), (3, 21, 64), (2, 7, 16)])
@pytest.mark.parametrize('cin', [1, 3])
def test_conv_first_layer(ops, cin, shape, relu):
    rng = np.random.default_rng(cin)
    n, h, w = shape
    hi = 256 if cin == 3 else 65536
    x = rng.integers(0, hi, (n, h, w, cin)).astype(np.float32)
    wt = (rng.standard_normal((3, 3, cin, 64)) * 0.05).astype(np.float32)
    b = rng.standard_normal(64).astype(np.float32)
    y = ops.Act(n, h, w, 64)
    ops.conv2d_first_fwd(_dev(x), _dev(wt), _dev(b), y, relu=relu)
    torch.cuda.synchronize()
    y32, _ = _conv_oracle(x, wt, b, relu, 3)
    ref = y32.permute(0, 2, 3, 1).numpy()
    got = y.interior().float().cpu().numpy()
    # fp32 math on both sides (different summation order) then one bf16 rounding
    np.testing.assert_allclose(got, ref, rtol=2 ** -8, atol=1e-5 * hi)
    raw = y.t.view(torch.int16)
    assert not (raw[:, 0].any() or raw[:, -1].any() or raw[:, :, 0].any() or raw[:, :, -1].any())   # halo stays zero
    if relu:
        assert not (raw < 0).any()              # no negative zeros either: the next layer's relu mask reads the sign bit


def test_conv_first_layer_fp32_exactness(ops):
    """W % 16 == 0 takes the MFMA form (three-way bf16 split of every fp32 operand): on operands whose products and
    sums are exact in fp32 it must reproduce the exact result bit for bit, and on random fp32 data it must round to
    the same bf16 as a float64 evaluation except where the float64 value sits within fp32 noise of a rounding boundary."""
    rng = np.random.default_rng(11)
    n, h, w = 2, 18, 48
    for cin in (1, 3):
        # 16-bit integers x 12-bit dyadic weights: every product needs 28 bits -> only the three-way split gets them all
        x = rng.integers(0, 65536, (n, h, w, cin)).astype(np.float32)
        wt = (rng.integers(-2048, 2048, (3, 3, cin, 64)) / 2.0 ** 14).astype(np.float32)
        b = rng.integers(-8, 8, 64).astype(np.float32)
        y = ops.Act(n, h, w, 64)
        ops.conv2d_first_fwd(_dev(x), _dev(wt), _dev(b), y, relu=False)
        torch.cuda.synchronize()
        xt = torch.from_numpy(x).double().permute(0, 3, 1, 2)
        wtt = torch.from_numpy(wt).double().permute(3, 2, 0, 1)
        exact = torch.nn.functional.conv2d(xt, wtt, torch.from_numpy(b).double(), padding=1).permute(0, 2, 3, 1)
        got = y.interior().cpu()
        # the float64 result rounded to bf16, except where it sits within the fp32 accumulation's own rounding
        # (2^-22 of the summed magnitudes) of a bf16 rounding boundary
        want = exact.float().to(torch.bfloat16)
        mag = torch.nn.functional.conv2d(xt.abs(), wtt.abs(), torch.from_numpy(b).double().abs(), padding=1).permute(0, 2, 3, 1)
        diff = got != want
        if diff.any():
            e = exact[diff]
            ulp = 2.0 ** (torch.floor(torch.log2(e.abs())) - 7)
            frac = torch.remainder(e / ulp, 1.0)
            assert ((frac - 0.5).abs() * ulp <= mag[diff] * 2.0 ** -22).all()
            assert ((got[diff].double() - e).abs() <= ulp).all()
        assert diff.float().mean() < 2e-3
        # random fp32 data
        x = (rng.standard_normal((n, h, w, cin)) * 100).astype(np.float32)
        wt = (rng.standard_normal((3, 3, cin, 64)) * 0.1).astype(np.float32)
        b = rng.standard_normal(64).astype(np.float32)
        ops.conv2d_first_fwd(_dev(x), _dev(wt), _dev(b), y, relu=False)
        torch.cuda.synchronize()
        exact = torch.nn.functional.conv2d(torch.from_numpy(x).double().permute(0, 3, 1, 2),
                                           torch.from_numpy(wt).double().permute(3, 2, 0, 1),
                                           torch.from_numpy(b).double(), padding=1).permute(0, 2, 3, 1)
        got = y.interior().cpu()
        mism = (got != exact.float().to(torch.bfloat16)).float().mean().item()
        # measured 6e-5 .. 7e-5, the same as an fp32 FMA chain (torch fp32 on the CPU: 4e-5 .. 8e-5); a two-way split
        # (2^-17 relative) would flip ~100x more roundings
        assert mism < 4e-4, mism
        assert (got.float() - exact.float()).abs().max() <= 2.0 ** -7 * exact.abs().max()


def test_upsample2x_relu_add(ops):
    rng = np.random.default_rng(5)
    n, h, w, c = 2, 5, 7, 64
    x = fo.round_bf16(rng.standard_normal((n, h, w, c)).astype(np.float32))
    res = fo.round_bf16(np.abs(rng.standard_normal((n, 2 * h, 2 * w, c))).astype(np.float32))
    y = ops.upsample2x_relu_add(ops.Act.from_dense(_dev(x)), ops.Act.from_dense(_dev(res)))
    torch.cuda.synchronize()
    xt = torch.from_numpy(x).permute(0, 3, 1, 2)
    up = fo.deconv_same(xt, fo.bilinear_kernel(4, c), 2, relu=True).permute(0, 2, 3, 1).numpy()
    ref = fo.round_bf16(up + res)
    got = y.interior().float().cpu().numpy()
    # the x2 weights (1/16, 3/16, 9/16) are exact; only the 4-term fp32 sum order differs
    np.testing.assert_allclose(got, ref, rtol=2 ** -7, atol=1e-6)
    assert (got != ref).mean() < 0.01


@pytest.mark.parametrize('C,U', [(12, 64), (14, 64), (20, 128)])
def test_decoder_head(ops, C, U):
    rng = np.random.default_rng(C)
    n, h, w = 2, 3, 5
    f = fo.round_bf16(np.abs(rng.standard_normal((n, h, w, U))).astype(np.float32))
    ws = (rng.standard_normal((U, C)) * 0.3).astype(np.float32)
    bs = rng.standard_normal(C).astype(np.float32)
    out = ops.decoder_head_fwd(ops.Act.from_dense(_dev(f)), _dev(ws), _dev(bs), C, want_score=True, want_prob=True)
    torch.cuda.synchronize()
    ft = torch.from_numpy(f).permute(0, 3, 1, 2)
    up = fo.deconv_same(ft, fo.bilinear_kernel(16, U), 8, relu=True)
    score = fo.conv2d_same(up, ws.reshape(1, 1, U, C), bs).permute(0, 2, 3, 1).numpy()
    got_score = out['score'].cpu().numpy()
    np.testing.assert_allclose(got_score, score, rtol=1e-4, atol=1e-4)          # stated logit tolerance
    # probabilities and labels are checked against the oracle fed the SAME (kernel) logits
    np.testing.assert_allclose(out['prob'].cpu().numpy(), fo.softmax(got_score), rtol=1e-5, atol=1e-7)
    assert np.array_equal(out['label'].cpu().numpy(), fo.argmax_last(fo.softmax(got_score)))
    assert out['label'].dtype == torch.int64


def test_softmax_argmax_bit_exact_labels_and_ties(ops):
    rng = np.random.default_rng(11)
    s = (rng.standard_normal((3, 17, 23, 12)) * 4).astype(np.float32)
    s[0, 0, 0] = 0.0                      # all-equal logits -> class 0
    s[0, 0, 1] = [1, 5, 5, 0, 5, 0, 0, 0, 0, 0, 0, 0]   # tie -> lowest index
    s[0, 0, 2] = -200.0
    s[0, 0, 2, 7] = 90.0                  # saturated softmax
    prob, label = ops.softmax_argmax(_dev(s))
    torch.cuda.synchronize()
    np.testing.assert_allclose(prob.cpu().numpy(), fo.softmax(s), rtol=1e-5, atol=1e-30)
    assert np.array_equal(label.cpu().numpy(), fo.argmax_last(fo.softmax(s)))
    assert label[0, 0, 0].item() == 0 and label[0, 0, 1].item() == 1 and label[0, 0, 2].item() == 7


def _notebook_mats(golden_dir):
    import os
    g = np.load(os.path.join(golden_dir, 'notebook_868.npz'))
    return [g['cm_rgb'].astype('float32').T, g['cm_depth'].astype('float32').T]


@pytest.mark.parametrize('prior', ['data', 'uniform', 0.5])
def test_bayes_fuse_matches_oracle_and_reference_lut(ops, golden_dir, prior):
    import os
    from modular_semantic_segmentation_amd import bayes_mix
    mats = _notebook_mats(golden_dir)
    rng = np.random.default_rng(0)
    la = rng.integers(0, 12, (2, 37, 41)).astype(np.int64)
    lb = rng.integers(0, 12, (2, 37, 41)).astype(np.int64)
    loglik, logprior = bayes_mix.bayes_tables(mats, prior)
    fused, score = ops.bayes_fuse([_dev(la), _dev(lb)], _dev(loglik), _dev(logprior), want_score=True)
    torch.cuda.synchronize()
    ref_score, _, _ = fu.bayes_fusion([la, lb], mats, prior)
    np.testing.assert_allclose(score.cpu().numpy(), ref_score, rtol=1e-6, atol=1e-5)
    assert np.array_equal(fused.cpu().numpy(), np.argmax(score.cpu().numpy(), -1))
    # against the reference's own decision matrix (golden): identical wherever the decision is not a near-tie
    name = {'data': 'data', 'uniform': 'uniform', 0.5: 'w0p5'}[prior]
    lut = np.load(os.path.join(golden_dir, 'bayes_lut.npz'))['lut_' + name]
    top2 = np.sort(ref_score, -1)[..., -2:]
    clear = (top2[..., 1] - top2[..., 0]) > 1e-4
    assert np.array_equal(fused.cpu().numpy()[clear], lut[la, lb][clear])
    fused_lut = ops.bayes_fuse_lut(_dev(la), _dev(lb), _dev(lut))
    torch.cuda.synchronize()
    assert np.array_equal(fused_lut.cpu().numpy(), lut[la, lb])


def test_dirichlet_fuse(ops):
    from modular_semantic_segmentation_amd import dirichlet_mix
    rng = np.random.default_rng(4)
    C = 12
    pa = fo.softmax((rng.standard_normal((2, 19, 21, C)) * 3).astype(np.float32))
    pb = fo.softmax((rng.standard_normal((2, 19, 21, C)) * 3).astype(np.float32))
    pa[0, 0, 0] = 0
    pa[0, 0, 0, 3] = 1.0          # exact zeros: exercises the 1e-20 guard
    A = [rng.uniform(0.3, 5.0, (C, C)).astype(np.float32) for _ in range(2)]
    counts = rng.integers(1, 1000, C)
    for prior_cfg, sigma in (('data', 1.0), ('uniform', 0.5), (0.3, 2.0)):
        am1, lognorm, logprior = dirichlet_mix.dirichlet_tables(A, counts, prior_cfg, sigma)
        fused, score = ops.dirichlet_fuse([_dev(pa), _dev(pb)], _dev(am1), _dev(lognorm), _dev(logprior), want_score=True)
        torch.cuda.synchronize()
        prior = fu.dirichlet_prior(counts, prior_cfg)
        ref = fu.dirichlet_fusion([fu.renormalise(pa), fu.renormalise(pb)], A, prior, sigma)
        np.testing.assert_allclose(score.cpu().numpy(), ref, rtol=2e-5, atol=2e-3)
        assert np.array_equal(fused.cpu().numpy(), np.argmax(score.cpu().numpy(), -1))
        top2 = np.sort(ref, -1)[..., -2:]
        clear = (top2[..., 1] - top2[..., 0]) > 1e-2
        assert np.array_equal(fused.cpu().numpy()[clear], np.argmax(ref, -1)[clear])


@pytest.mark.parametrize('shape', [(1, 1, 1), (2, 19, 21), (1, 37, 53)])
def test_dirichlet_fuse_packed_form_equals_scalar_form(ops, shape):
    """Two experts, 12 classes: the fusion kernel runs on packed fp32 with the table transposed (dirichlet_fuse_pk_kernel);
    XV_DIRICHLET_FUSE_PK=0 keeps the scalar form the test above compares with the oracle.  Same IEEE operations in the same
    order: scores and labels must be equal bit for bit, for pixel counts that are not multiples of the workgroup."""
    from modular_semantic_segmentation_amd import dirichlet_mix
    rng = np.random.default_rng(14)
    C = 12
    pa = fo.softmax((rng.standard_normal(shape + (C,)) * 3).astype(np.float32))
    pb = fo.softmax((rng.standard_normal(shape + (C,)) * 3).astype(np.float32))
    pa[0, 0, 0] = 0
    pa[0, 0, 0, 5] = 1.0
    A = [rng.uniform(0.3, 5.0, (C, C)).astype(np.float32) for _ in range(2)]
    am1, lognorm, logprior = dirichlet_mix.dirichlet_tables(A, rng.integers(1, 1000, C), 'data', 1.0)
    args = ([_dev(pa), _dev(pb)], _dev(am1), _dev(lognorm), _dev(logprior))
    old = os.environ.get('XV_DIRICHLET_FUSE_PK')
    try:
        os.environ['XV_DIRICHLET_FUSE_PK'] = '0'
        ref = ops.dirichlet_fuse(*args, want_score=True)
        ref_l, _ = ops.dirichlet_fuse(*args)
        os.environ['XV_DIRICHLET_FUSE_PK'] = '1'
        got = ops.dirichlet_fuse(*args, want_score=True)
        got_l, _ = ops.dirichlet_fuse(*args)
    finally:
        if old is None:
            os.environ.pop('XV_DIRICHLET_FUSE_PK', None)
        else:
            os.environ['XV_DIRICHLET_FUSE_PK'] = old
    assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])
    assert torch.equal(got_l, ref_l) and torch.equal(got_l, ref[0])


def test_average_fuse(ops):
    rng = np.random.default_rng(8)
    pa = fo.softmax(rng.standard_normal((1, 9, 9, 14)).astype(np.float32))
    pb = fo.softmax(rng.standard_normal((1, 9, 9, 14)).astype(np.float32))
    fused = ops.average_fuse([_dev(pa), _dev(pb)])
    torch.cuda.synchronize()
    assert np.array_equal(fused.cpu().numpy(), np.argmax((pa + pb) / np.float32(2), -1))


def test_confusion_matrix_and_suffstats(ops):
    rng = np.random.default_rng(9)
    C = 12
    lab = rng.integers(-1, C, (3, 33, 47)).astype(np.int32)
    pred = rng.integers(0, C, (3, 33, 47)).astype(np.int64)
    cm = torch.zeros((C, C), dtype=torch.int64, device='cuda')
    ops.confusion_matrix(_dev(lab), _dev(pred), cm)
    ops.confusion_matrix(_dev(lab), _dev(pred), cm)          # accumulates
    torch.cuda.synchronize()
    assert np.array_equal(cm.cpu().numpy(), 2 * fu.confusion_matrix(lab, pred, C))
    p = fo.softmax((rng.standard_normal((3, 33, 47, C)) * 2).astype(np.float32))
    S = torch.zeros((C, C), dtype=torch.float64, device='cuda')
    cnt = torch.zeros(C, dtype=torch.int64, device='cuda')
    ops.dirichlet_suffstats(_dev(p), _dev(lab), S, cnt)
    torch.cuda.synchronize()
    Sref, nref = fu.sufficient_statistics(p, lab, C)
    assert np.array_equal(cnt.cpu().numpy(), nref)
    np.testing.assert_allclose(S.cpu().numpy(), Sref, rtol=1e-6, atol=1e-4)


@pytest.mark.parametrize('k,s,n,h,w,cin,cout', [(4, 2, 2, 6, 10, 64, 64), (16, 8, 1, 5, 7, 64, 64), (4, 2, 1, 3, 4, 128, 64)])
def test_dense_transposed_conv_exact_on_integers(ops, k, s, n, h, w, cin, cout):
    """xv_deconv_dense_fwd (the fallback for deconv kernels that are NOT the bilinear constant, custom_layers.py:71-121):
    tf.layers.conv2d_transpose(k, strides=s, 'same', no bias) as one 3x3 MFMA conv over the s*s output phases +
    depth-to-space, with the batch-norm affine, relu and residual add behind it -- bit-exact on integer operands
    against torch's conv_transpose2d with [TF1] padding (k - s) // 2."""
    from modular_semantic_segmentation_amd.custom_layers import dense_deconv_as_conv3x3
    rng = np.random.default_rng(k * 100 + cin)
    x = rng.integers(-2, 3, (n, h, w, cin)).astype(np.float32)
    wt = rng.integers(-1, 2, (k, k, cout, cin)).astype(np.float32)         # [kh, kw, out, in]
    res = rng.integers(-3, 4, (n, h * s, w * s, cout)).astype(np.float32)
    scale = rng.integers(1, 3, cout).astype(np.float32)
    shift = rng.integers(-2, 3, cout).astype(np.float32)
    wp = ops.pack_conv_weights(_dev(dense_deconv_as_conv3x3(wt, s)))
    zb = torch.zeros(s * s * cout, device='cuda')
    xa = ops.Act.from_dense(_dev(x))
    y, ws = ops.deconv_dense_fwd(xa, wp, zb, s, cout, scale=_dev(scale), shift=_dev(shift), residual=ops.Act.from_dense(_dev(res)),
                                 relu=True)
    torch.cuda.synchronize()
    xt = torch.from_numpy(x).permute(0, 3, 1, 2).contiguous()
    raw = fo.deconv_same(xt, wt, s)                                         # exact integers
    raw16 = fo.round_bf16(raw)                                              # the phase map is stored as bf16
    ref = torch.relu(raw16 * torch.from_numpy(scale).view(1, -1, 1, 1) + torch.from_numpy(shift).view(1, -1, 1, 1)) + \
        torch.from_numpy(res).permute(0, 3, 1, 2)
    assert np.array_equal(y.interior().float().cpu().numpy(), fo.round_bf16(ref).permute(0, 2, 3, 1).numpy())
    for edge in (y.t[:, 0], y.t[:, -1], y.t[:, :, 0], y.t[:, :, -1]):
        assert not edge.any()
    y2, _ = ops.deconv_dense_fwd(xa, wp, zb, s, cout, relu=False, workspace=ws)          # plain form, workspace reused
    torch.cuda.synchronize()
    assert np.array_equal(y2.interior().float().cpu().numpy(), raw16.permute(0, 2, 3, 1).numpy())


@pytest.mark.gpu
def test_pack_table_matches_single_layer_packers(ops):
    """The one-launch re-pack of a training step (xv_pack_conv_weights_multi, 8 channels per thread) writes the same bits
    as the per-layer entry points (one element per thread): forward and data-gradient images, 3x3 (two images) and 1x1."""
    g = torch.Generator().manual_seed(5)
    shapes = [(3, 64, 64), (3, 64, 128), (3, 128, 64), (3, 256, 512), (1, 512, 64), (1, 64, 128)]
    entries, want = [], []
    for k, cin, cout in shapes:
        w = torch.randn(k, k, cin, cout, generator=g).cuda()
        fwd = torch.zeros(ops.packed_weight_elems(k, cin, cout), dtype=torch.bfloat16, device='cuda')
        bwd = torch.zeros(ops.packed_weight_elems(k, cout, cin), dtype=torch.bfloat16, device='cuda')
        entries.append((w, fwd, bwd))
        want.append((ops.pack_conv_weights(w), ops.pack_conv_weights_dgrad(w)))
    fwd_only = torch.zeros_like(entries[1][1])
    entries.append((entries[1][0], fwd_only, None))
    ops.PackTable(entries, 'cuda').run()
    torch.cuda.synchronize()
    for (w, fwd, bwd), (wf, wb) in zip(entries, want):
        assert torch.equal(fwd.view(torch.int16), wf.view(torch.int16)), tuple(w.shape)
        assert torch.equal(bwd.view(torch.int16), wb.view(torch.int16)), tuple(w.shape)
    assert torch.equal(fwd_only.view(torch.int16), want[1][0].view(torch.int16))


def test_conv2d_chunk_groups_and_split_form_agree_bit_for_bit(ops):
    """Generation 5 on the shapes only it serves (24x48 conv5 maps, 512 channels: conv_col_dma.hip round 5): the sum over
    input channels goes by chunk GROUPS, in registers for a full chip (MODE 1) and as one work item per (tile, group) plus a
    reduction launch where the tiles fill less than half the CUs (MODE 2, the batch-1 latency form).  Both add the same
    groups in the same order: exact on integers against the oracle, and -- on random operands -- the split launch of ONE image
    equals that image inside a batch of 12 (unsplit) bit for bit."""
    from modular_semantic_segmentation_amd import _lib
    lib = _lib.lib()
    h, w, cin, cout = 24, 48, 512, 512
    assert lib.xv_conv2d_split_workspace_bytes(1, h, w, cin, cout) == 24 * 4 * 24 * 16 * 64 * 4     # 24 tiles x 4 groups
    assert lib.xv_conv2d_split_workspace_bytes(2, h, w, cin, cout) == 2 * 24 * 4 * 24 * 16 * 64 * 4
    assert lib.xv_conv2d_split_workspace_bytes(12, h, w, cin, cout) == 0                            # 288 tiles: whole tiles
    assert lib.xv_conv2d_split_workspace_bytes(1, 48, 96, cin, cout) == 0                           # tiles in 16x32: no groups
    assert lib.xv_conv2d_split_workspace_bytes(1, h, w, 128, cout) == 0                             # four chunks: no groups
    rng = np.random.default_rng(12)
    arena = {}
    for exact in (True, False):
        if exact:
            x = rng.integers(-2, 3, (12, h, w, cin)).astype(np.float32)
            wt = rng.integers(-1, 2, (3, 3, cin, cout)).astype(np.float32)
            b = rng.integers(-3, 4, cout).astype(np.float32)
        else:
            x = rng.standard_normal((12, h, w, cin)).astype(np.float32)
            wt = (rng.standard_normal((3, 3, cin, cout)) / np.sqrt(9 * cin)).astype(np.float32)
            b = rng.standard_normal(cout).astype(np.float32)
        wp, bd = ops.pack_conv_weights(_dev(wt)), _dev(b)
        xa = ops.Act.from_dense(_dev(x))
        y_batch, _ = ops.conv2d_fwd(xa, wp, bd, 3, relu=True)                       # 288 tiles: chunk groups in registers
        for n in (1, 2):
            xs = ops.Act.from_dense(_dev(x[:n]))
            ws = ops.split_workspace(xs, cout, arena)
            assert ws is not None
            y_split, _ = ops.conv2d_fwd(xs, wp, bd, 3, relu=True, split_ws=ws)      # the split form
            y_plain, _ = ops.conv2d_fwd(xs, wp, bd, 3, relu=True)                   # no workspace: groups in registers
            torch.cuda.synchronize()
            assert torch.equal(y_split.t, y_plain.t)
            assert torch.equal(y_split.interior(), y_batch.interior()[:n])
            full = y_split.t.float().cpu().numpy()
            assert np.all(full[:, 0] == 0) and np.all(full[:, -1] == 0) and np.all(full[:, :, 0] == 0) and np.all(full[:, :, -1] == 0)
        _, ref = _conv_oracle(x[:2], wt, b, True, 3)
        got = y_batch.interior()[:2].float().cpu().numpy()
        if exact:
            assert np.array_equal(got, ref)
        else:
            assert np.abs(got - ref).max() <= 2 ** -7 * np.abs(ref).max()           # bf16 output rounding
    # no relu: negative outputs survive the reduction launch
    xs = ops.Act.from_dense(_dev(x[:1]))
    y0, _ = ops.conv2d_fwd(xs, wp, bd, 3, relu=False, split_ws=ops.split_workspace(xs, cout, arena))
    y1, _ = ops.conv2d_fwd(xs, wp, bd, 3, relu=False)
    torch.cuda.synchronize()
    assert torch.equal(y0.t, y1.t) and bool((y0.interior() < 0).any())
