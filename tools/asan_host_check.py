#!/usr/bin/env python3
"""Host half of libxview_hip.so under AddressSanitizer + UndefinedBehaviorSanitizer (CPU box; no GPU needed).

    make -C modular_semantic_segmentation_amd/csrc asan && python tools/asan_host_check.py

`make asan` compiles the HOST pass of every source with -fsanitize=address,undefined (device code as usual; GPU-side
sanitizers are not available on this pool) into tools/build/libxview_hip_asan.so.  This script loads it through the same
ctypes table as the product (_lib.SIGNATURES) -- in a child process with the ASan runtime preloaded -- and drives everything
the library does on the host without a device:

  1. every symbol include/xview_hip.h declares is exported;
  2. the argument / shape checks of the entry points (null pointers, misaligned pointers, wrong dtypes, shapes that do not
     tile) return XV_EINVAL / XV_ESHAPE / XV_EWORKSPACE before anything is launched;
  3. the size calculators (packed weights, every *_workspace_bytes) and the tile chooser (xv_conv2d_choose_cfg = pick_cfg)
     over every layer of every BASELINE.json shape, every flag combination, plus a fuzz over random and extreme dimensions
     (int overflow in tile counts / byte sizes is what UBSan is here for);
  4. the launch-geometry arithmetic of the conv entry points: with fake (never dereferenced) device pointers the call runs
     the whole host path up to the kernel launch, which fails with a HIP error on a box without a GPU.

Any sanitizer report aborts the child (-fno-sanitize-recover, halt_on_error); exit code 0 = clean."""
import ctypes
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASAN_LIB = os.path.join(ROOT, 'tools', 'build', 'libxview_hip_asan.so')


def child():
    sys.path.insert(0, ROOT)
    import numpy as np
    from modular_semantic_segmentation_amd import _lib
    _lib.LIB_PATH = ASAN_LIB
    h = _lib.lib()                                          # (1) AttributeError if an export is missing
    print('exports: %d symbols' % len(_lib.SIGNATURES))
    EINVAL, ESHAPE, EWS = -1, -2, -3
    checks = 0

    def expect(code, want, what):
        nonlocal checks
        checks += 1
        assert code in (want if isinstance(want, tuple) else (want,)), '%s returned %d, expected %s' % (what, code, want)

    # (2) argument checks
    act = _lib.xv_act
    fake = 0x7f0000000000                                    # 16-byte aligned, never dereferenced on the host
    a0 = act(None, 1, 16, 16, 64, 0, 0)
    expect(h.xv_softmax_argmax(None, 10, 12, None, None, None), EINVAL, 'xv_softmax_argmax(null)')
    expect(h.xv_bayes_fuse_lut(None, None, None, 12, 10, None, None), EINVAL, 'xv_bayes_fuse_lut(null)')
    expect(h.xv_maxpool2x2_fwd(a0, a0, None), EINVAL, 'xv_maxpool2x2_fwd(null data)')
    x = act(fake, 2, 24, 40, 128, 0, 0)
    y = act(fake + (1 << 30), 2, 24, 40, 256, 0, 0)
    expect(h.xv_conv2d_fwd_cfg(x, None, None, y, None, 3, 1, -1, None), EINVAL, 'conv: null weights')
    expect(h.xv_conv2d_fwd_cfg(x, fake, fake, act(fake, 2, 24, 41, 256, 0, 0), None, 3, 1, -1, None), ESHAPE, 'conv: shape mismatch')
    expect(h.xv_conv2d_fwd_cfg(act(fake + 8, 2, 24, 40, 128, 0, 0), fake, fake, y, None, 3, 1, -1, None), EINVAL, 'conv: misaligned')
    expect(h.xv_conv2d_fwd_cfg(x, fake, fake, y, None, 5, 1, -1, None), ESHAPE, 'conv: k = 5')
    expect(h.xv_conv2d_fwd_cfg(x, fake, fake, y, None, 3, 1, 999, None), EINVAL, 'conv: cfg out of range')
    expect(h.xv_conv2d_fwd_cfg(act(fake, 2, 24, 40, 100, 0, 0), fake, fake, y, None, 3, 1, -1, None), ESHAPE, 'conv: cin % 64')
    expect(h.xv_conv2d_fwd_cfg(x, fake, fake, y, act(fake, 2, 12, 20, 128, 0, 0), 3, 1, -1, None), ESHAPE, 'conv: pooled channels')
    expect(h.xv_conv2d_fwd_cfg(x, fake, fake, y, None, 3, 1, 27, None), ESHAPE, 'conv: cfg 27 on a map that does not tile')
    expect(h.xv_conv2d_fwd_cfg(act(fake, 2, 24, 40, 128, 1, 300), fake, fake, y, None, 3, 1, -1, None), ESHAPE, 'conv: fp8 exponent')
    expect(h.xv_conv2d_fwd_ws(x, fake, fake, y, None, 3, 1, -1, fake, 16, None), EWS, 'conv: workspace too small')
    # the two-model launch: null twin, mismatched twins, a shape the generation-4 / 5 kernels do not take
    expect(h.xv_conv2d_fwd_pair(x, fake, fake, y, None, None, fake, fake, y, None, 1, None), EINVAL, 'pair: null twin')
    expect(h.xv_conv2d_fwd_pair(x, fake, fake, y, None, act(fake, 3, 24, 40, 128, 0, 0), fake, fake, y, None, 1, None), ESHAPE,
           'pair: mismatched twins')
    expect(h.xv_conv2d_fwd_pair(x, fake, fake, y, None, x, fake, fake, y, None, 1, None), ESHAPE, 'pair: 24x40 does not tile')
    expect(h.xv_conv2d_fwd_pair(x, fake, fake, y, None, x, fake, fake, act(fake, 2, 24, 40, 256, 1, 0), None, 1, None), EINVAL,
           'pair: fp8 twin')
    expect(h.xv_maxpool2x2_fwd(act(fake, 1, 16, 16, 64, 1, 0), act(fake, 1, 8, 8, 64, 1, 0), None), EINVAL, 'pool: fp8 descriptor')

    # (3) size calculators and the tile chooser
    layers = [('conv1_2', 1, 64, 64), ('conv2_1', 2, 64, 128), ('conv2_2', 2, 128, 128), ('conv3_1', 4, 128, 256),
              ('conv3_2', 4, 256, 256), ('conv4_1', 8, 256, 512), ('conv4_2', 8, 512, 512), ('conv5_1', 16, 512, 512)]
    shapes = [(1, 256, 512), (16, 384, 768), (1, 384, 768), (2, 512, 1024), (4, 1024, 2048), (1, 1024, 2048), (3, 250, 330)]
    ncfg = h.xv_conv2d_num_cfgs()
    seen = {}
    for n, H, W in shapes:
        for name, s, cin, cout in layers:
            hh, ww = max(H // s, 1), max(W // s, 1)
            for flags in range(8):
                for din, dout in ((0, 0), (0, 1), (1, 1)):
                    c = h.xv_conv2d_choose_cfg(n, hh, ww, cin, cout, 3, din, dout, flags)
                    checks += 1
                    assert c in (ESHAPE,) or 0 <= c < ncfg, (name, n, hh, ww, flags, din, dout, c)
                    if (flags, din, dout) == (0, 0, 0):
                        seen[(n, H, W, name)] = c
            assert h.xv_packed_weight_bytes(3, cin, cout) == 3 * 9 * cin * cout * 2
            assert h.xv_packed_weight_bytes_f8(3, cin, cout) == 256 + 2 * 9 * cin * cout
            ws = h.xv_conv2d_bwd_filter_workspace_bytes(n, hh, ww, cin, cout, 3)
            assert ws >= 9 * cin * cout * 4, (name, ws)
            checks += 3
        for C in (12, 14, 19):
            assert h.xv_decoder_head_workspace_bytes(n, H // 8, W // 8, C) > 0
            assert h.xv_decoder_head_bwd_workspace_bytes(n, H // 8, W // 8, C) > 0
        assert h.xv_conv2d_first_bwd_filter_workspace_bytes(n, H, W, 3) > 0
        assert h.xv_deconv_dense_workspace_bytes(n, H // 16, W // 16, 64, 2) == n * (H // 16 + 2) * (W // 16 + 2) * 4 * 64 * 2
        for k, cin, cout in ((1, 512, 64), (1, 256, 128), (1, 2048, 512)):
            c = h.xv_conv2d_choose_cfg(n, H // 8, W // 8, cin, cout, k, 0, 0, 0)
            assert 0 <= c < ncfg
        checks += 9
    # the headline shapes take the round-4 kernels
    assert seen[(16, 384, 768, 'conv3_2')] == 26 and seen[(16, 384, 768, 'conv1_2')] == 26, seen
    assert seen[(16, 384, 768, 'conv5_1')] == 27 and seen[(4, 1024, 2048, 'conv5_1')] == 26, seen
    assert h.xv_conv2d_streamk_workspace_bytes() > 0
    rng = np.random.default_rng(0)
    dims = [1, 2, 3, 7, 16, 24, 47, 48, 96, 384, 768, 1024, 2048, 4096, 32768, 65535, 1 << 20, (1 << 31) - 1]
    chans = [64, 128, 192, 256, 512, 1024, 4096, 1 << 20, 100, 0, -64]
    for _ in range(20000):
        n, hh, ww = (int(rng.choice(dims)) for _ in range(3))
        cin, cout = int(rng.choice(chans)), int(rng.choice(chans))
        k = int(rng.choice([1, 3, 2]))
        c = h.xv_conv2d_choose_cfg(n, hh, ww, cin, cout, k, int(rng.integers(0, 3)), int(rng.integers(0, 2)), int(rng.integers(0, 8)))
        assert c in (EINVAL, ESHAPE) or 0 <= c < ncfg
        h.xv_packed_weight_bytes(k, cin, cout)
        h.xv_packed_weight_bytes_f8(k, cin, cout)
        h.xv_conv2d_bwd_filter_workspace_bytes(n, hh, ww, cin, cout, k)
        h.xv_decoder_head_workspace_bytes(n, hh, ww, int(rng.integers(-1, 40)))
        h.xv_decoder_head_bwd_workspace_bytes(n, hh, ww, int(rng.integers(-1, 40)))
        h.xv_deconv_dense_workspace_bytes(n, hh, ww, cout, int(rng.integers(-1, 9)))
        h.xv_conv2d_first_bwd_filter_workspace_bytes(n, hh, ww, int(rng.integers(-1, 6)))
        h.xv_conv2d_route_bytes(n, hh, ww, cout)
        h.xv_score_dense_bwd_workspace_bytes(n, hh, ww)
        h.xv_conv2d_split_workspace_bytes(n, hh, ww, cin, cout)
        h.xv_softmax_ce_dense_workspace_bytes(n * hh * ww)
        h.xv_upsample_raw_bwd_workspace_bytes(n, hh, ww, cout)
        checks += 13

    # (4) launch geometry: the whole host path of the conv entry points up to the launch (which fails: no device here)
    import torch
    if not torch.cuda.is_available():
        launched = 0
        for n, H, W in shapes[:5]:
            for name, s, cin, cout in layers:
                hh, ww = H // s, W // s
                xa, ya = act(fake, n, hh, ww, cin, 0, 0), act(fake + (1 << 32), n, hh, ww, cout, 0, 0)
                pool = act(fake + (1 << 33), n, hh // 2, ww // 2, cout, 0, 0) if (hh % 2 == 0 and ww % 2 == 0) else None
                for cfg in range(-1, ncfg):
                    for p in (None, pool):
                        rc = h.xv_conv2d_fwd_cfg(xa, fake, fake, ya, p, 3, 1, cfg, None)
                        assert rc != 0, 'a launch succeeded without a GPU?'
                        launched += 1
                rc = h.xv_conv2d_bwd_data(ya, fake, fake, xa, xa, xa, 3, None)
                assert rc != 0
                for p in (None, pool):
                    rc = h.xv_conv2d_fwd_pair(xa, fake, fake, ya, p, xa, fake, fake, ya, p, 1, None)
                    assert rc != 0
                    launched += 1
        checks += launched
        print('launch geometry: %d conv entry calls ran to the (failing) launch' % launched)
    print('asan/ubsan host check: %d checks, no sanitizer report' % checks)


def main():
    if os.environ.get('XV_ASAN_CHILD') == '1':
        return child()
    if not os.path.exists(ASAN_LIB):
        sys.exit('build it first: make -C modular_semantic_segmentation_amd/csrc asan')
    rt = sorted(glob.glob('/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so'))
    if not rt:
        sys.exit('ASan runtime not found under /opt/rocm/lib/llvm')
    env = dict(os.environ, XV_ASAN_CHILD='1', LD_PRELOAD=rt[-1],
               ASAN_OPTIONS='detect_leaks=0:halt_on_error=1:abort_on_error=0:detect_odr_violation=0',
               UBSAN_OPTIONS='halt_on_error=1:print_stacktrace=1')
    rc = subprocess.call([sys.executable, os.path.abspath(__file__)], env=env)
    sys.exit(rc)


if __name__ == '__main__':
    main()
