"""Stream-K tail of the generation-2 3x3 conv kernel (conv_dma_kernel; replaces tf.layers.conv2d of
simple_fcn.py:39-79 like the plain kernel): when the persistent grid's last round of tiles is incomplete, its (tile,
32-channel chunk) items are dealt out over all workgroups, partial sums meet in fp32 slabs and the workgroup that
arrives last adds them in a fixed order.  Checked here through the C ABI (xv_conv2d_fwd_ws / xv_conv2d_bwd_data_ws):

  * bit-exact against the oracle on integer operands (sums of small integers are exact in any order), for shapes whose
    tail splits tiles 2 .. 16 ways, with partial edge tiles, the fused pool, pooled-only outputs and the 24x16 tile;
  * bit-identical run to run on random operands (the slab order is fixed, whoever arrives last) and within one bf16
    rounding of the unsplit kernel's result;
  * the arrival counters are zero again after every launch; a workspace that is too small is refused;
  * the data gradient (mask / addend epilogue) through the same path.
"""
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
    y = fo.conv2d_same(torch.from_numpy(x).permute(0, 3, 1, 2).contiguous(), wt, b, relu=relu)
    return y, fo.round_bf16(y).permute(0, 2, 3, 1).numpy()


def _counters(ws):
    return ws[:4096].view(torch.int32)


# n, h, w, cin, cout, pool, cfg     (256 CUs: tiles of 16x32 (cfg 17) / 24x16 (cfg 22) x 64 output channels)
CASES = [
    (1, 48, 96, 512, 512, False, 17),     # conv4_x at one image: 72 tiles on 256 CUs, 16 chunks each -> ~4.5 items per CU
    (1, 24, 48, 512, 512, False, 22),     # conv5_x at one image: 24 tiles, each split ~11 ways
    (1, 96, 192, 256, 256, True, 17),     # conv3_3 at one image with the fused pool: 144 tiles
    (2, 48, 96, 256, 512, False, 17),     # 144 tiles, 8 chunks
    (3, 40, 72, 128, 192, False, 17),     # partial edge tiles (40 = 2.5 x 16, 72 = 2.25 x 32), 81 tiles, 4 chunks
    (5, 32, 64, 128, 640, True, 17),      # 200 tiles + pool
    (9, 32, 64, 64, 256, False, 17),      # 144 tiles of TWO chunks (the resident-weights layers' shape)
    (1, 384, 768, 64, 64, True, 17),      # conv1_2 at one image: 2 rounds + 8 tiles per XCD group = 16 tail items for 32
    #                                       workgroups (some get none)
    (1, 64, 160, 64, 64, False, 17),      # 20 tiles of two chunks: 40 items on 256 CUs
    (20, 32, 64, 256, 128, False, 17),    # 160 tiles < 256
    (21, 32, 64, 128, 192, False, 17),    # 252 tiles: almost a full round, tail not worth splitting on some XCD groups
    (27, 32, 64, 128, 192, True, 17),     # 324 tiles: one full round + a tail of 68
    (1, 48, 96, 256, 256, False, -1),     # the library's own choice of tile
]


@pytest.mark.parametrize('n,h,w,cin,cout,pool,cfg', CASES)
def test_streamk_conv_exact_on_integers(ops, n, h, w, cin, cout, pool, cfg):
    rng = np.random.default_rng(abs(hash((n, h, w, cin, cout))) % 2**32)
    x = rng.integers(-2, 3, (n, h, w, cin)).astype(np.float32)
    wt = rng.integers(-1, 2, (3, 3, cin, cout)).astype(np.float32)
    b = rng.integers(-3, 4, cout).astype(np.float32)
    xa, wp, bd = ops.Act.from_dense(_dev(x)), ops.pack_conv_weights(_dev(wt)), _dev(b)
    ws = ops.streamk_workspace('cuda')
    q = ops.Act(n, h // 2, w // 2, cout) if pool else None
    y, _ = ops.conv2d_fwd(xa, wp, bd, 3, relu=True, pooled=q, cfg=cfg, workspace=ws)
    torch.cuda.synchronize()
    y32, ref = _oracle(x, wt, b, True)
    got = y.interior().float().cpu().numpy()
    bad = np.argwhere(got != ref)
    assert bad.size == 0, 'first mismatches (n,y,x,c): %s got %s want %s' % (
        bad[:5].tolist(), got[tuple(bad[:5].T)], ref[tuple(bad[:5].T)])
    for edge in (y.t[:, 0], y.t[:, -1], y.t[:, :, 0], y.t[:, :, -1]):
        assert not edge.any()                                       # the zero border is untouched
    assert not _counters(ws).any()                                  # every arrival counter is back to zero
    if pool:
        refq = fo.round_bf16(fo.maxpool2(y32)).permute(0, 2, 3, 1).numpy()
        assert np.array_equal(q.interior().float().cpu().numpy(), refq)
        q3 = ops.Act(n, h // 2, w // 2, cout)
        ops.conv2d_fwd(xa, wp, bd, 3, relu=True, pooled=q3, write_y=False, cfg=cfg, workspace=ws)   # pooled-only launch
        torch.cuda.synchronize()
        assert torch.equal(q3.t, q.t) and not _counters(ws).any()
    # the same workspace again, and the kernel without one: identical bits
    y2, _ = ops.conv2d_fwd(xa, wp, bd, 3, relu=True, cfg=cfg, workspace=ws)
    y0, _ = ops.conv2d_fwd(xa, wp, bd, 3, relu=True, cfg=cfg)
    torch.cuda.synchronize()
    assert torch.equal(y2.t, y.t) and torch.equal(y0.t, y.t)


@pytest.mark.parametrize('n,h,w,cin,cout,cfg', [(1, 48, 96, 512, 512, 17), (1, 24, 48, 512, 512, 22),
                                                (3, 40, 72, 256, 192, 17), (16, 24, 48, 512, 512, 22)])
def test_streamk_conv_is_reproducible_on_random_operands(ops, n, h, w, cin, cout, cfg):
    """Random operands: the split tiles' fp32 sums are grouped differently from the unsplit kernel's (so single bf16
    roundings may differ from it), but identically from run to run -- the slabs are added in workgroup order whichever
    workgroup arrives last."""
    g = torch.Generator(device='cuda').manual_seed(n * h + cin)
    x = torch.randn((n, h, w, cin), device='cuda', generator=g)
    wt = torch.randn((3, 3, cin, cout), device='cuda', generator=g) * (1.0 / (9 * cin) ** 0.5)
    b = torch.randn(cout, device='cuda', generator=g)
    xa, wp = ops.Act.from_dense(x), ops.pack_conv_weights(wt)
    ws = ops.streamk_workspace('cuda')
    first = None
    for rep in range(6):
        if rep == 3:          # other work between the launches shifts the arrival order
            torch.randn((4096, 4096), device='cuda') @ torch.randn((4096, 4096), device='cuda')
        y, _ = ops.conv2d_fwd(xa, wp, b, 3, relu=False, cfg=cfg, workspace=ws)
        torch.cuda.synchronize()
        if first is None:
            first = y.t.clone()
        assert torch.equal(y.t, first), 'run %d differs' % rep
    assert not _counters(ws).any()
    plain, _ = ops.conv2d_fwd(xa, wp, b, 3, relu=False, cfg=cfg)
    torch.cuda.synchronize()
    a, p = first.float(), plain.t.float()
    # one bf16 rounding step at most (2^-8 relative), on a small fraction of the values
    assert (a - p).abs().max() <= 2.0 ** -7 * p.abs().max()
    assert (a != p).float().mean() < 0.05
    ref = fo.conv2d_same(x.cpu().bfloat16().float().permute(0, 3, 1, 2).contiguous(), wt.cpu().bfloat16().float().numpy(),
                         b.cpu().numpy(), relu=False).permute(0, 2, 3, 1)
    np.testing.assert_allclose(y.interior().float().cpu().numpy(), ref.numpy(), rtol=2.0 ** -7, atol=2.0 ** -7 * float(ref.abs().max()))


def test_streamk_data_gradient_with_mask_and_addend(ops):
    """xv_conv2d_bwd_data_ws: Conv2DBackpropInput + AddN + ReluGrad in the epilogue of a split tile."""
    rng = np.random.default_rng(5)
    n, h, w, cin, cout = 1, 48, 96, 256, 512               # dy has 512 channels (16 chunks), dx 256
    dy = rng.integers(-2, 3, (n, h, w, cout)).astype(np.float32)
    wt = rng.integers(-1, 2, (3, 3, cin, cout)).astype(np.float32)
    ref_act = rng.integers(-1, 2, (n, h, w, cin)).astype(np.float32)
    add = rng.integers(-3, 4, (n, h, w, cin)).astype(np.float32)
    wd = ops.pack_conv_weights_dgrad(_dev(wt))
    zeros = torch.zeros(cin, device='cuda')
    dya, ra, aa = ops.Act.from_dense(_dev(dy)), ops.Act.from_dense(_dev(ref_act)), ops.Act.from_dense(_dev(add))
    ws = ops.streamk_workspace('cuda')
    dx = ops.conv2d_bwd_data(dya, wd, zeros, ops.Act(n, h, w, cin), 3, relu_ref=ra, addend=aa, workspace=ws)
    dx0 = ops.conv2d_bwd_data(dya, wd, zeros, ops.Act(n, h, w, cin), 3, relu_ref=ra, addend=aa)
    torch.cuda.synchronize()
    wt_t = torch.from_numpy(wt).permute(3, 2, 0, 1).contiguous()
    ref = torch.nn.grad.conv2d_input((n, cin, h, w), wt_t, torch.from_numpy(dy).permute(0, 3, 1, 2).contiguous(), padding=1)
    ref = (ref.permute(0, 2, 3, 1) + torch.from_numpy(add)) * (torch.from_numpy(ref_act) > 0)
    assert torch.equal(dx.interior().float().cpu(), fo.round_bf16(ref))
    assert torch.equal(dx.t, dx0.t) and not _counters(ws).any()


def test_streamk_workspace_contract(ops):
    from modular_semantic_segmentation_amd import _lib
    need = _lib.lib().xv_conv2d_streamk_workspace_bytes()
    assert need >= 4096 + 2 * 256 * 96 * 1024
    x = ops.Act(1, 16, 32, 64)
    wp = ops.pack_conv_weights(torch.zeros((3, 3, 64, 64), device='cuda'))
    small = torch.zeros(need - 16, dtype=torch.uint8, device='cuda')
    with pytest.raises(_lib.XvError, match='XV_EWORKSPACE'):
        ops.conv2d_fwd(x, wp, torch.zeros(64, device='cuda'), 3, workspace=small)
    # kernels other than generation 2 ignore the workspace (1x1 conv, first-generation tiles)
    ws = ops.streamk_workspace('cuda')
    y1, _ = ops.conv2d_fwd(x, ops.pack_conv_weights(torch.ones((1, 1, 64, 64), device='cuda')), torch.zeros(64, device='cuda'), 1,
                           workspace=ws)
    torch.cuda.synchronize()
    assert not y1.interior().any() and not _counters(ws).any()


def test_streamk_engine_matches_plain_engine(ops):
    """The FCN expert with and without the workspace (XV_DMA_NO_STREAMK is the library's A/B switch; here the engine's
    workspace is simply withheld): reproducible, and logits within compounded bf16 rounding of the plain engine's at one image, where every layer
    from conv2 on has a split tail."""
    from modular_semantic_segmentation_amd.fcn import FcnEngine
    w = fo.init_fcn_weights('rgb', 3, 64, 12, seed=1, bias_scale=0.02)
    w['rgb/conv1_1/kernel'] *= 0.02
    for k in w:
        if k.endswith('/kernel') and 'upscore' not in k and 'conv1_1' not in k:
            w[k] *= 1.6
    x = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (1, 384, 768, 3)).astype(np.float32)).cuda()
    eng = FcnEngine('rgb', 3, 64, 12, w, streamk=True)
    a = eng.forward(x, want=('score', 'label'))
    sa, la = a['score'].clone(), a['label'].clone()
    again = eng.forward(x, want=('score', 'label'))
    torch.cuda.synchronize()
    assert torch.equal(again['score'], sa) and torch.equal(again['label'], la)        # reproducible
    assert not _counters(eng._sk()).any()
    assert FcnEngine('rgb', 3, 64, 12, w)._sk() is None           # off by default: results independent of the batch
    eng.streamk = False
    b = eng.forward(x, want=('score', 'label'))
    torch.cuda.synchronize()
    scale = float(b['score'].abs().max())
    # (different fp32 groupings flip single bf16 roundings, which compound through 13 layers of a random-init net like
    # any other rounding difference: the same 2 % of the logit scale as the end-to-end oracle comparison)
    assert float((sa - b['score']).abs().max()) < 2e-2 * scale
    assert float((la == b['label']).float().mean()) > 0.97
