// Filter gradient of conv2d (3x3 'same' / 1x1) on bf16 MFMA for gfx950.
//
// Replaces the Conv2DBackpropFilter ops that tf.train.*Optimizer.minimize adds for every
// tf.layers.conv2d of the FCN (base_model.py:153-162 over simple_fcn.py:39-79):
//     dW[tap][cin][cout] = sum_{n,y,x} X[n, y+dy-1, x+dx-1, cin] * dY[n, y, x, cout]
//
// GEMM view: M = cin, N = cout, K = pixels.  Both MFMA operands are needed K(=pixel)-major while
// NHWC memory is channel-major, so the fragments are fetched with the gfx950 transposing LDS read
// `ds_read_b64_tr_b16` from the same swizzled [pixel][64 ch] LDS images the forward kernel uses:
// an 8x32-pixel dY tile and the 10x34 halo patch of X (staged once, re-read by all 9 taps).
// One K-step = 32 consecutive pixels of one image row; the K index <-> pixel map inside a step is
// permuted (x = 16*g1 + 8*h + 4*g0 + q for lane group g = 2*g1+g0, element j = 4*h+q) so that every
// 32-lane half of a transposing read touches 8 consecutive pixels = all 64 banks once.
// A workgroup owns a (64 cin) x (64 cout) block of dW for all taps and a slice of the pixel tiles
// (split-K); wave w accumulates cin rows [16w, 16w+16) x 64 cout x 9 taps = 144 fp32 registers and
// finally adds them into dW with fp32 global atomics (run-to-run summation order is not fixed).
#include "xv_common.h"

// Diagnostic builds only (tools/wgrad_exp.sh; wrong results): XV_WGRAD_EXP bit 0 = no LDS-DMA after a workgroup's first tile,
// bit 1 = no per-tile barrier, bit 2 = fragments read for the first row of a tile only.  Which of the three the matrix pipe
// waits for is read off the launch time of each build.
#ifndef XV_WGRAD_EXP
#define XV_WGRAD_EXP 0
#endif

#ifndef XV_WGRAD_UNROLL
#define XV_WGRAD_UNROLL 8  // rows of a tile fully unrolled: +30 % over the rolled loop (profiles/r1_conv_tune_*.txt)
#endif

namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4;

struct WgradArgs {
  const __bf16* x;
  const __bf16* dy;
  float* dw;
  float* db;  // may be null; accumulated by the workgroups of input-channel block 0 from their dY tiles
  float* slab;  // may be null: per-split partial dW slabs [splits][taps][Cin][Cout] (plain stores, reduced by a
                // second kernel in a fixed order) instead of fp32 atomics straight into dw
  float* bslab;  // with slab: per-split partial bias gradients [splits][Cout], reduced by the same second kernel
  int N, H, W, Cin, Cout;
  int tiles_x, tiles_y, n_ptiles, splits;
};

__device__ inline bf16x8 tr_read2(const char* smem, int addr_lo, int addr_hi) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(smem + addr_lo));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(smem + addr_hi));
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

template <int KS>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(WgradArgs a) {
  constexpr int TH = 8, TW = 32;
  constexpr int HALO = (KS == 3) ? 1 : 0;
  constexpr int HH = TH + 2 * HALO, HW = TW + 2 * HALO;
  constexpr int NPIX = HH * HW;
  constexpr int X_BYTES = ((NPIX * 128 + 255) / 256) * 256;
  constexpr int NTAPS = KS * KS;
  constexpr int X_ITERS = (NPIX * 8 + 255) / 256;
  constexpr int D_ITERS = TH * TW * 8 / 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const Xs = smem;
  char* const Ds = smem + X_BYTES;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout;
  const int Wp = W + 2;

  const int split = blockIdx.x % a.splits;
  const int pair = blockIdx.x / a.splits;
  const int n_co = Cout >> 6;
  const int co0 = (pair % n_co) << 6;
  const int ci0 = (pair / n_co) << 6;
  const int per = (a.n_ptiles + a.splits - 1) / a.splits;
  const int t_begin = split * per;
  const int t_end = t_begin + per < a.n_ptiles ? t_begin + per : a.n_ptiles;

  // lane roles inside a transposing read (see header): group g = lane>>4, q = row (pixel), p = 4-channel piece
  const int g = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
  const int xk = 16 * (g >> 1) + 4 * (g & 1) + q;  // pixel column of element j = q (h = 0); h = 1 adds 8
  int xbase[KS];                                   // X patch: row 0, column xk + dx, this wave's 16 cin
#pragma unroll
  for (int dx = 0; dx < KS; ++dx) {
    const int c = xk + dx;
    xbase[dx] = c * 128 + (xv_swz(c, wave * 2 + (p >> 1)) << 4) + (p & 1) * 8;
  }
  int dbase[4];  // dY tile: row 0, column xk, cout tile n
#pragma unroll
  for (int n = 0; n < 4; ++n) dbase[n] = X_BYTES + xk * 128 + (xv_swz(xk, n * 2 + (p >> 1)) << 4) + (p & 1) * 8;

  f32x4 acc[NTAPS][4];
#pragma unroll
  for (int t = 0; t < NTAPS; ++t)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[t][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  // BiasAddGrad rides along: thread (cout = tid & 63, pixel quarter = tid >> 6) sums the staged dY tile
  const bool do_bias = a.db != nullptr && ci0 == 0;
  float bsum = 0.f;

  for (int t = t_begin; t < t_end; ++t) {
    const int tx = t % a.tiles_x;
    int r = t / a.tiles_x;
    const int ty = r % a.tiles_y;
    const int n = r / a.tiles_y;
    const int y0 = ty * TH, x0 = tx * TW;
    const __bf16* ximg = a.x + (int64_t)n * (H + 2) * Wp * Cin + ci0;
    const __bf16* dimg = a.dy + (int64_t)n * (H + 2) * Wp * Cout + co0;
    __syncthreads();  // previous tile's fragments are consumed
#pragma unroll
    for (int half = 0; half < 2; ++half) {  // two halves: 144 accumulators leave room for ~24 staging registers
      constexpr int XH = (X_ITERS + 1) / 2;
      u32x4 v[XH];
#pragma unroll
      for (int i2 = 0; i2 < XH; ++i2) {
        const int it = half * XH + i2;
        int idx = tid + it * 256;
        idx = idx < NPIX * 8 ? idx : NPIX * 8 - 1;
        const int pp = idx >> 3, s = idx & 7;
        const int hy = pp / HW, hx = pp - hy * HW;
        int yy = y0 + hy + (1 - HALO), xx = x0 + hx + (1 - HALO);  // padded coords, clamped onto the zero border
        yy = yy < H + 1 ? yy : H + 1;
        xx = xx < W + 1 ? xx : W + 1;
        v[i2] = *reinterpret_cast<const u32x4*>(ximg + ((int64_t)yy * Wp + xx) * Cin + s * 8);
      }
#pragma unroll
      for (int i2 = 0; i2 < XH; ++i2) {
        const int it = half * XH + i2;
        const int idx = tid + it * 256;
        const int pp = idx >> 3, s = idx & 7;
        const int hx = pp % HW;
        if (it < X_ITERS && idx < NPIX * 8) *reinterpret_cast<u32x4*>(Xs + pp * 128 + (xv_swz(hx, s) << 4)) = v[i2];
      }
    }
    {
      u32x4 v[D_ITERS];
#pragma unroll
      for (int it = 0; it < D_ITERS; ++it) {
        const int idx = tid + it * 256;
        const int pp = idx >> 3, s = idx & 7;
        const int py = pp / TW, px = pp - py * TW;
        int yy = y0 + py + 1, xx = x0 + px + 1;
        yy = yy < H + 1 ? yy : H + 1;  // rows / columns past the image read the zero border: no contribution
        xx = xx < W + 1 ? xx : W + 1;
        v[it] = *reinterpret_cast<const u32x4*>(dimg + ((int64_t)yy * Wp + xx) * Cout + s * 8);
      }
#pragma unroll
      for (int it = 0; it < D_ITERS; ++it) {
        const int idx = tid + it * 256;
        const int pp = idx >> 3, s = idx & 7;
        const int px = pp % TW;
        *reinterpret_cast<u32x4*>(Ds + pp * 128 + (xv_swz(px, s) << 4)) = v[it];
      }
    }
    __syncthreads();
    if (do_bias) {
      const int co = tid & 63, qtr = tid >> 6;
      const int slot = co >> 3, e = co & 7;
#pragma unroll 4
      for (int pp = qtr * 64; pp < qtr * 64 + 64; ++pp) {
        const int px = pp & (TW - 1);
        bsum += (float)*reinterpret_cast<const __bf16*>(Ds + pp * 128 + (xv_swz(px, slot) << 4) + e * 2);
      }
    }
#pragma unroll 1
    for (int y = 0; y < TH; ++y) {  // rolled: 144 accumulators leave no room for cross-row hoisting
      const int yd = y * (TW * 128), yx = y * (HW * 128);
      bf16x8 bfr[4];
#pragma unroll
      for (int n4 = 0; n4 < 4; ++n4) bfr[n4] = tr_read2(smem, dbase[n4] + yd, dbase[n4] + yd + 8 * 128);
#pragma unroll
      for (int tap = 0; tap < NTAPS; ++tap) {
        const int dy = (KS == 3) ? tap / 3 : 0;
        const int dx = (KS == 3) ? tap % 3 : 0;
        const bf16x8 afr = tr_read2(smem, xbase[dx] + yx + dy * (HW * 128), xbase[dx] + yx + dy * (HW * 128) + 8 * 128);
#pragma unroll
        for (int n4 = 0; n4 < 4; ++n4)
          acc[tap][n4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr, bfr[n4], acc[tap][n4], 0, 0, 0);
      }
    }
  }

  if (do_bias) {
    if (a.bslab != nullptr) {
      // deterministic: the four pixel quarters of a channel meet in LDS and are added in a fixed order
      __syncthreads();  // (the last tile's fragments are consumed)
      float* red = reinterpret_cast<float*>(smem);
      red[tid] = bsum;
      __syncthreads();
      if (tid < 64) a.bslab[(int64_t)split * Cout + co0 + tid] = ((red[tid] + red[64 + tid]) + red[128 + tid]) + red[192 + tid];
    } else {
      atomicAdd(a.db + co0 + (tid & 63), bsum);
    }
  }
  // accumulator (row = cin = 4*(lane>>4) + r, col = cout = lane & 15) -> dW[tap][cin][cout] (HWIO)
  const int cin = ci0 + wave * 16 + g * 4;
  const int cout = co0 + li;
  float* const out = a.slab ? a.slab + (int64_t)split * NTAPS * Cin * Cout : a.dw;
#pragma unroll
  for (int tap = 0; tap < NTAPS; ++tap)
#pragma unroll
    for (int n4 = 0; n4 < 4; ++n4) {
      float* dst = out + ((int64_t)tap * Cin + cin) * Cout + cout + n4 * 16;
      if (a.slab) {
        dst[0] = acc[tap][n4].x;
        dst[Cout] = acc[tap][n4].y;
        dst[2 * Cout] = acc[tap][n4].z;
        dst[3 * Cout] = acc[tap][n4].w;
      } else {
        atomicAdd(dst, acc[tap][n4].x);
        atomicAdd(dst + Cout, acc[tap][n4].y);
        atomicAdd(dst + 2 * Cout, acc[tap][n4].z);
        atomicAdd(dst + 3 * Cout, acc[tap][n4].w);
      }
    }
}

// ---- version 2: LDS-DMA double buffering -----------------------------------------------------------
// One 8-wave workgroup per CU.  While the matrix cores work on pixel tile t out of LDS buffer t&1, the
// next tile's X halo patch and dY tile stream HBM/L2 -> LDS directly (`global_load_lds_dwordx4`, no
// VGPR staging; the swizzle is applied through the per-lane SOURCE address because an LDS-DMA
// instruction writes 1 KB linearly), so a tile costs exactly one barrier and the staging latency is
// covered by a whole tile of MFMA work.  Wave w accumulates cin rows [16*(w&3), +16) x 64 cout for the
// taps of group w>>2 (taps 0-4 / 5-8): 80 / 64 accumulator registers.
template <int KS>
__global__ __launch_bounds__(512, 2) void conv_wgrad_dma_kernel(WgradArgs a) {
  constexpr int TH = 8, TW = 32;
  constexpr int HALO = (KS == 3) ? 1 : 0;
  constexpr int HH = TH + 2 * HALO, HW = TW + 2 * HALO;
  constexpr int NPIX = HH * HW;
  constexpr int X_BYTES = ((NPIX * 128 + 1023) / 1024) * 1024;  // whole 1-KB DMA pieces
  constexpr int D_BYTES = TH * TW * 128;
  constexpr int BUF = X_BYTES + D_BYTES;
  constexpr int NTAPS = KS * KS;
  constexpr int T0 = (NTAPS + 1) / 2;        // taps of group 0: [0, T0), group 1: [T0, NTAPS)
  constexpr int XI = X_BYTES / 1024, DI = D_BYTES / 1024;  // DMA instructions per tile
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cb = wave & 3, tg = wave >> 2;
  const int H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout;
  const int Wp = W + 2;

  const int split = blockIdx.x % a.splits;
  const int pair = blockIdx.x / a.splits;
  const int n_co = Cout >> 6;
  const int co0 = (pair % n_co) << 6;
  const int ci0 = (pair / n_co) << 6;
  const int per = (a.n_ptiles + a.splits - 1) / a.splits;
  const int t_begin = split * per;
  const int t_end = t_begin + per < a.n_ptiles ? t_begin + per : a.n_ptiles;

  const int g = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
  const int xk = 16 * (g >> 1) + 4 * (g & 1) + q;
  int xbase[KS];
#pragma unroll
  for (int dx = 0; dx < KS; ++dx) {
    const int c = xk + dx;
    xbase[dx] = c * 128 + (xv_swz(c, cb * 2 + (p >> 1)) << 4) + (p & 1) * 8;
  }
  int dbase[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) dbase[n] = X_BYTES + xk * 128 + (xv_swz(xk, n * 2 + (p >> 1)) << 4) + (p & 1) * 8;

  // LDS-DMA of a pixel tile: 1-KB piece i of the tile (X patch pieces 0 .. XI-1, then the dY pieces) is moved by wave
  // i % 8, i.e. this wave moves pieces wave + 8 j, j < NJ.  Their per-lane source offsets relative to the tile's origin
  // do not depend on the tile (except on the image's right / bottom edge, where coordinates are clamped onto the zero
  // border), so they are computed ONCE -- the ~20 integer instructions per piece used to run, for all ~10 pieces of a
  // wave in one burst, at the start of every tile, in front of its first MFMA.  The pieces of tile t+1 are issued one per
  // image row of tile t (the CU's vector-memory pipe takes a piece every ~25 cycles and a wave whose DMA does not fit its
  // queue stalls in order, MFMAs included: the forward kernel's lesson).
  constexpr int NJ = (XI + DI + 7) / 8;
  int voff[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int i = wave + 8 * j;
    if (i < XI) {
      int idx = i * 64 + lane;  // 16-byte piece of the patch image: pixel idx>>3, PHYSICAL slot idx&7
      idx = idx < NPIX * 8 ? idx : NPIX * 8 - 1;
      const int pp = idx >> 3, ps = idx & 7;
      const int hy = pp / HW, hx = pp - hy * HW;
      voff[j] = ((hy * Wp + hx) * Cin + xv_swz(hx, ps) * 8) * 2;  // logical slot stored at this physical slot (involution)
    } else {
      const int idx = (i - XI) * 64 + lane;
      const int pp = idx >> 3, ps = idx & 7;
      const int py = pp / TW, px = pp - py * TW;
      voff[j] = ((py * Wp + px) * Cout + xv_swz(px, ps) * 8) * 2;
    }
  }
  struct TileSrc {
    const __bf16 *xt, *dt;  // origins of the tile's X patch / dY tile (padded coordinates)
    int y0, x0;
    bool edge;
  };
  auto tile_src = [&](int t) {
    const int tx = t % a.tiles_x;
    int r = t / a.tiles_x;
    const int ty = r % a.tiles_y;
    const int n = r / a.tiles_y;
    TileSrc ts;
    ts.y0 = ty * TH;
    ts.x0 = tx * TW;
    ts.edge = ts.y0 + TH > H || ts.x0 + TW > W;
    const __bf16* ximg = a.x + (int64_t)n * (H + 2) * Wp * Cin + ci0;
    const __bf16* dimg = a.dy + (int64_t)n * (H + 2) * Wp * Cout + co0;
    ts.xt = ximg + ((int64_t)(ts.y0 + 1 - HALO) * Wp + (ts.x0 + 1 - HALO)) * Cin;
    ts.dt = dimg + ((int64_t)(ts.y0 + 1) * Wp + (ts.x0 + 1)) * Cout;
    return ts;
  };
  // piece wave + 8 j of a tile into buffer b (in assembly, SGPR base + 32-bit lane offset: the builtin makes hipcc model
  // a FLAT access, after which every LDS wait it inserts is lgkmcnt(0) instead of a counted one)
  auto stage_piece = [&](const TileSrc& ts, int j, int b) {
    const int i = wave + 8 * j;
    if (i >= XI + DI) return;
    int off = voff[j];
    const bool isx = i < XI;
    if (ts.edge) {
      // right / bottom edge of the image: coordinates past it are clamped onto the zero border (rare path: the offset is
      // re-derived from the lane number here instead of living in registers)
      int ln = lane;
      asm volatile("" : "+v"(ln));
      if (isx) {
        int idx = i * 64 + ln;
        idx = idx < NPIX * 8 ? idx : NPIX * 8 - 1;
        const int pp = idx >> 3, ps = idx & 7;
        const int hy = pp / HW, hx = pp - hy * HW;
        int yy = ts.y0 + hy + (1 - HALO), xx = ts.x0 + hx + (1 - HALO);
        yy = (yy < H + 1 ? yy : H + 1) - (ts.y0 + 1 - HALO);
        xx = (xx < W + 1 ? xx : W + 1) - (ts.x0 + 1 - HALO);
        off = ((yy * Wp + xx) * Cin + xv_swz(hx, ps) * 8) * 2;
      } else {
        const int idx = (i - XI) * 64 + ln;
        const int pp = idx >> 3, ps = idx & 7;
        const int py = pp / TW, px = pp - py * TW;
        int yy = ts.y0 + py + 1, xx = ts.x0 + px + 1;
        yy = (yy < H + 1 ? yy : H + 1) - (ts.y0 + 1);
        xx = (xx < W + 1 ? xx : W + 1) - (ts.x0 + 1);
        off = ((yy * Wp + xx) * Cout + xv_swz(px, ps) * 8) * 2;
      }
    }
    const int lds = __builtin_amdgcn_readfirstlane(b * BUF + (isx ? i * 1024 : X_BYTES + (i - XI) * 1024));
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds), "v"(off), "s"(isx ? ts.xt : ts.dt) : "memory");
  };
  auto stage = [&](int t, int b) {  // a whole tile at once (the first tile of a workgroup)
    const TileSrc ts = tile_src(t);
#pragma unroll
    for (int j = 0; j < NJ; ++j) stage_piece(ts, j, b);
  };

  constexpr int NT = (T0 > NTAPS - T0) ? T0 : NTAPS - T0;  // accumulator tap slots per wave
  f32x4 acc[NT][4];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[t][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool do_bias = a.db != nullptr && ci0 == 0;
  float bsum = 0.f;

  // (the LDS-DMA is issued in assembly, invisible to the compiler's wait insertion, and __syncthreads() waits for LDS
  // operations only: the vector-memory counter is drained by hand before every barrier that publishes a staged tile)
  if (t_begin < t_end) stage(t_begin, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int t = t_begin; t < t_end; ++t) {
    const int b = (t - t_begin) & 1;
    const bool more = t + 1 < t_end;
    const TileSrc nxt = tile_src(more ? t + 1 : t);
    const char* buf = smem + b * BUF;
    if (do_bias) {
      const int co = tid & 63, part = tid >> 6;  // 8 parts x 32 pixels
      const int slot = co >> 3, e = co & 7;
#pragma unroll 4
      for (int pp = part * 32; pp < part * 32 + 32; ++pp) {
        const int px = pp & (TW - 1);
        bsum += (float)*reinterpret_cast<const __bf16*>(buf + X_BYTES + pp * 128 + (xv_swz(px, slot) << 4) + e * 2);
      }
    }
    // Row y's fragments (4 dY tiles + one X fragment per tap of this wave's group) are requested one row AHEAD into a
    // second register set, so a row is: [request row y+1] [wait for row y] [its 16-20 MFMAs in one burst at raised
    // priority].  Two waves share a SIMD: the burst priority makes them alternate row by row instead of interleaving
    // MFMA by MFMA (the forward kernel's scheme), and while one bursts the other's requests are in flight.
    auto frag_row = [&](int y, bf16x8 (&bfr)[4], bf16x8 (&afr)[NT]) {
      const int yd = y * (TW * 128), yx = y * (HW * 128);
#pragma unroll
      for (int n4 = 0; n4 < 4; ++n4) bfr[n4] = tr_read2(buf, dbase[n4] + yd, dbase[n4] + yd + 8 * 128);
#pragma unroll
      for (int ts = 0; ts < NT; ++ts) {
        // (both groups' taps with compile-time indices: a runtime index into xbase[] would put it in scratch)
        const int t0 = ts, t1 = T0 + ts < NTAPS ? T0 + ts : NTAPS - 1;
        const int a0 = xbase[(KS == 3) ? t0 % 3 : 0] + ((KS == 3) ? t0 / 3 : 0) * (HW * 128);
        const int a1 = xbase[(KS == 3) ? t1 % 3 : 0] + ((KS == 3) ? t1 / 3 : 0) * (HW * 128);
        const int ad = (tg == 0 ? a0 : a1) + yx;
        if ((tg == 0 ? ts : T0 + ts) < NTAPS) afr[ts] = tr_read2(buf, ad, ad + 8 * 128);
      }
    };
    bf16x8 bfr[2][4], afr[2][NT];
    frag_row(0, bfr[0], afr[0]);
#pragma unroll
    for (int y = 0; y < TH; ++y) {
      // the next tile's pieces: two per row over the first rows, so that the last of them has the remaining rows' MFMAs to
      // land behind (issued one per row up to the last row, the tile ended waiting for its youngest piece)
      if (more && 2 * y < NJ && !(XV_WGRAD_EXP & 1)) {
        stage_piece(nxt, 2 * y, b ^ 1);
        if (2 * y + 1 < NJ) stage_piece(nxt, 2 * y + 1, b ^ 1);
      }
      if (y + 1 < TH && !(XV_WGRAD_EXP & 4)) frag_row(y + 1, bfr[(y + 1) & 1], afr[(y + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ts = 0; ts < NT; ++ts) {
        const int tap = tg == 0 ? ts : T0 + ts;
        if (tap < NTAPS) {
#pragma unroll
          for (int n4 = 0; n4 < 4; ++n4) {
            acc[ts][n4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[(XV_WGRAD_EXP & 4) ? 0 : (y & 1)][ts],
                                                                  bfr[(XV_WGRAD_EXP & 4) ? 0 : (y & 1)][n4], acc[ts][n4], 0, 0, 0);
            if (ts == 0 && n4 == 0) __builtin_amdgcn_s_setprio(2);
          }
        }
      }
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (!(XV_WGRAD_EXP & 2)) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();  // DMA of tile t+1 has landed and every wave is done with buffer b
    }
  }

  if (do_bias) {
    if (a.bslab != nullptr) {
      // deterministic: the eight pixel parts of a channel meet in LDS (both tile buffers are consumed: the loop ends
      // with a barrier) and are added in a fixed order
      float* red = reinterpret_cast<float*>(smem);
      red[tid] = bsum;
      __syncthreads();
      if (tid < 64) {
        float t = red[tid];
#pragma unroll
        for (int part = 1; part < 8; ++part) t += red[part * 64 + tid];
        a.bslab[(int64_t)split * Cout + co0 + tid] = t;
      }
    } else {
      atomicAdd(a.db + co0 + (tid & 63), bsum);
    }
  }
  const int cin = ci0 + cb * 16 + g * 4;
  const int cout = co0 + li;
  float* const out = a.slab ? a.slab + (int64_t)split * NTAPS * Cin * Cout : a.dw;
#pragma unroll
  for (int ts = 0; ts < NT; ++ts) {
    const int tap = tg == 0 ? ts : T0 + ts;
    if (tap < (tg == 0 ? T0 : NTAPS)) {
#pragma unroll
      for (int n4 = 0; n4 < 4; ++n4) {
        float* dst = out + ((int64_t)tap * Cin + cin) * Cout + cout + n4 * 16;
        if (a.slab) {
          dst[0] = acc[ts][n4].x;
          dst[Cout] = acc[ts][n4].y;
          dst[2 * Cout] = acc[ts][n4].z;
          dst[3 * Cout] = acc[ts][n4].w;
        } else {
          atomicAdd(dst, acc[ts][n4].x);
          atomicAdd(dst + Cout, acc[ts][n4].y);
          atomicAdd(dst + 2 * Cout, acc[ts][n4].z);
          atomicAdd(dst + 3 * Cout, acc[ts][n4].w);
        }
      }
    }
  }
}

// ---- version 3 (round 6): loader waves + halo-row-major fragments ---------------------------------------------------------
// What version 2 waits for was measured with parts of its tile loop switched off (tools/wgrad_exp.sh, profiles/r6_wgrad_exp.txt;
// conv3_2, 16 images, one box): 299 us as it is, 250 without the LDS-DMA of the next tile, 264 without the per-row fragment
// reads, 177 without both (the per-tile barrier alone: nothing); matrix pipe 47 % busy AT FULL CLOCK (rocprofv3 counters,
// profiles/r6_wgrad_counters.json: no LDS bank conflict, LDS array 24 % busy) -- not power, not bandwidth: every one of its
// eight waves carries MFMAs, 18 transposing reads per 20 MFMAs and the DMA instructions in ONE in-order stream, and a wave
// stalled in the vector-memory queue or behind its reads issues no MFMA.
//   * ROLES.  Waves 0-3 (one per SIMD) compute and issue nothing but LDS reads and MFMAs; waves 4-7 (their SIMD partners) move
//     the next tile by LDS-DMA a whole tile ahead, add up the bias gradient from the staged dY tile and otherwise sleep at the
//     barrier.  A stalled DMA instruction stalls a loader.
//   * FRAGMENT REUSE.  Compute wave cb owns input channels 16 cb .. 16 cb + 15 x 64 output channels x ALL nine taps (144
//     accumulators, as version 1).  The X fragment of halo row r and column offset kx serves the taps (ky, kx) of the three
//     output rows y = r - ky: the wave walks the ten HALO rows, reads three X fragments per row and keeps the dY fragments of
//     three output rows in a ring -- 30 + 32 fragment reads per tile and wave where version 2 makes 8 x 18 per tile in each of
//     its eight waves (248 KB of LDS reads per tile instead of 576).
//   * Order inside halo row r: request X(r + 1); MFMAs of ky = 2 (output row r - 2, whose ring slot is then free); request
//     dY(r + 1) into that slot; MFMAs of ky = 1, ky = 0 -- 24 to 36 MFMAs (400-600 cycles) between a request and its first use.
// Same tile geometry, LDS images, split-K slabs and reduction kernels as version 2; per accumulator the pixels are added in
// tile order, rows in order (another order than version 2's: bitwise reproducible against itself, equal on integers).
__global__ __launch_bounds__(512, 2) void conv_wgrad_lw_kernel(WgradArgs a) {
  constexpr int TH = 8, TW = 32, HH = TH + 2, HW = TW + 2, NPIX = HH * HW;
  constexpr int X_BYTES = ((NPIX * 128 + 1023) / 1024) * 1024;
  constexpr int D_BYTES = TH * TW * 128;
  constexpr int BUF = X_BYTES + D_BYTES;
  constexpr int XI = X_BYTES / 1024, DI = D_BYTES / 1024;
  constexpr int NL = 4;                                  // loader waves
  constexpr int NJ = (XI + DI + NL - 1) / NL;            // DMA pieces per loader wave and tile
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout;
  const int Wp = W + 2;
  const int split = blockIdx.x % a.splits;
  const int pair = blockIdx.x / a.splits;
  const int n_co = Cout >> 6;
  const int co0 = (pair % n_co) << 6;
  const int ci0 = (pair / n_co) << 6;
  const int per = (a.n_ptiles + a.splits - 1) / a.splits;
  const int t_begin = split * per;
  const int t_end = t_begin + per < a.n_ptiles ? t_begin + per : a.n_ptiles;
  const bool do_bias = a.db != nullptr && ci0 == 0;
  float* const out = a.slab ? a.slab + (int64_t)split * 9 * Cin * Cout : a.dw;

  if (wave >= NL) {
    // ================================================ loader waves ================================================
    const int lw = wave - NL;
    int voff[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int i = lw + NL * j;
      if (i < XI) {
        int idx = i * 64 + lane;  // 16-byte piece of the patch image: pixel idx >> 3, PHYSICAL slot idx & 7
        idx = idx < NPIX * 8 ? idx : NPIX * 8 - 1;
        const int pp = idx >> 3, ps = idx & 7;
        const int hy = pp / HW, hx = pp - hy * HW;
        voff[j] = ((hy * Wp + hx) * Cin + xv_swz(hx, ps) * 8) * 2;
      } else {
        const int idx = (i - XI) * 64 + lane;
        const int pp = idx >> 3, ps = idx & 7;
        const int py = pp / TW, px = pp - py * TW;
        voff[j] = ((py * Wp + px) * Cout + xv_swz(px, ps) * 8) * 2;
      }
    }
    auto stage = [&](int t, int b) {
      const int tx = t % a.tiles_x;
      int r = t / a.tiles_x;
      const int ty = r % a.tiles_y;
      const int n = r / a.tiles_y;
      const int y0 = ty * TH, x0 = tx * TW;
      const bool edge = y0 + TH > H || x0 + TW > W;
      const __bf16* xt = a.x + (int64_t)n * (H + 2) * Wp * Cin + ci0 + ((int64_t)y0 * Wp + x0) * Cin;
      const __bf16* dt = a.dy + (int64_t)n * (H + 2) * Wp * Cout + co0 + ((int64_t)(y0 + 1) * Wp + (x0 + 1)) * Cout;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int i = lw + NL * j;
        if (i >= XI + DI) continue;
        int off = voff[j];
        const bool isx = i < XI;
        if (edge) {  // right / bottom edge of the image: coordinates past it are clamped onto the zero border (rare path)
          int ln = lane;
          asm volatile("" : "+v"(ln));
          if (isx) {
            int idx = i * 64 + ln;
            idx = idx < NPIX * 8 ? idx : NPIX * 8 - 1;
            const int pp = idx >> 3, ps = idx & 7;
            const int hy = pp / HW, hx = pp - hy * HW;
            int yy = y0 + hy, xx = x0 + hx;
            yy = (yy < H + 1 ? yy : H + 1) - y0;
            xx = (xx < W + 1 ? xx : W + 1) - x0;
            off = ((yy * Wp + xx) * Cin + xv_swz(hx, ps) * 8) * 2;
          } else {
            const int idx = (i - XI) * 64 + ln;
            const int pp = idx >> 3, ps = idx & 7;
            const int py = pp / TW, px = pp - py * TW;
            int yy = y0 + py + 1, xx = x0 + px + 1;
            yy = (yy < H + 1 ? yy : H + 1) - (y0 + 1);
            xx = (xx < W + 1 ? xx : W + 1) - (x0 + 1);
            off = ((yy * Wp + xx) * Cout + xv_swz(px, ps) * 8) * 2;
          }
        }
        const int lds = __builtin_amdgcn_readfirstlane(b * BUF + (isx ? i * 1024 : X_BYTES + (i - XI) * 1024));
        asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds), "v"(off), "s"(isx ? xt : dt) : "memory");
      }
    };
    float bsum = 0.f;
    if (t_begin < t_end) stage(t_begin, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int t = t_begin; t < t_end; ++t) {
      const int b = (t - t_begin) & 1;
      if (t + 1 < t_end) stage(t + 1, b ^ 1);  // (buffer b ^ 1 was released by the barrier that ended tile t - 1)
      if (do_bias) {
        const char* buf = smem + b * BUF + X_BYTES;
        const int lt = tid - NL * 64;
        const int co = lt & 63, part = lt >> 6;  // 4 parts x 64 pixels
        const int slot = co >> 3, e = co & 7;
#pragma unroll 4
        for (int pp = part * 64; pp < part * 64 + 64; ++pp) {
          const int px = pp & (TW - 1);
          bsum += (float)*reinterpret_cast<const __bf16*>(buf + pp * 128 + (xv_swz(px, slot) << 4) + e * 2);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // tile t + 1 has landed; the compute waves are done with buffer b
    }
    if (do_bias) {
      const int lt = tid - NL * 64;
      if (a.bslab != nullptr) {
        // deterministic: the four pixel parts of a channel meet in LDS (both tile buffers are consumed) in a fixed order
        float* red = reinterpret_cast<float*>(smem);
        red[lt] = bsum;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (lt < 64) a.bslab[(int64_t)split * Cout + co0 + lt] = ((red[lt] + red[64 + lt]) + red[128 + lt]) + red[192 + lt];
      } else {
        atomicAdd(a.db + co0 + (lt & 63), bsum);
        __builtin_amdgcn_s_barrier();
      }
    } else {
      __builtin_amdgcn_s_barrier();
    }
    return;
  }

  // ================================================== compute waves ==================================================
  const int cb = wave;
  const int g = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
  const int xk = 16 * (g >> 1) + 4 * (g & 1) + q;
  int xbase[3];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) {
    const int c = xk + dx;
    xbase[dx] = c * 128 + (xv_swz(c, cb * 2 + (p >> 1)) << 4) + (p & 1) * 8;
  }
  int dbase[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) dbase[n] = X_BYTES + xk * 128 + (xv_swz(xk, n * 2 + (p >> 1)) << 4) + (p & 1) * 8;

  f32x4 acc[9][4];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[t][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  __builtin_amdgcn_s_barrier();  // (the loaders' first tile)
  __builtin_amdgcn_s_setprio(2);
  // The fragment registers live across tiles: the tile's ONE barrier sits in front of its LAST halo row -- by then every read
  // of this buffer is in registers (the row's own fragments were requested a row earlier) and the loaders have seen the next
  // tile land -- and the first fragments of the next tile are requested right behind it, under the last row's 12 MFMAs: a
  // tile starts with its operands in registers instead of an LDS round trip behind a barrier.
  bf16x8 xf[2][3], df[3][4];
  auto req_x = [&](const char* buf, int r, bf16x8 (&dst)[3]) {
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) dst[kx] = tr_read2(buf, xbase[kx] + r * (HW * 128), xbase[kx] + r * (HW * 128) + 8 * 128);
  };
  auto req_d = [&](const char* buf, int y, bf16x8 (&dst)[4]) {
#pragma unroll
    for (int n4 = 0; n4 < 4; ++n4) dst[n4] = tr_read2(buf, dbase[n4] + y * (TW * 128), dbase[n4] + y * (TW * 128) + 8 * 128);
  };
  if (t_begin < t_end) {
    req_x(smem, 0, xf[0]);
    req_d(smem, 0, df[0]);
  }
  for (int t = t_begin; t < t_end; ++t) {
    const char* buf = smem + ((t - t_begin) & 1) * BUF;
    const char* nbuf = smem + ((t + 1 - t_begin) & 1) * BUF;
#pragma unroll
    for (int r = 0; r < HH; ++r) {
      if (r + 1 < HH) req_x(buf, r + 1, xf[(r + 1) & 1]);
      if (r == HH - 1) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (requested a row ago: landed)
        __builtin_amdgcn_s_barrier();  // every compute wave holds its last fragments of this buffer; the next tile has landed
        if (t + 1 < t_end) {
          req_x(nbuf, 0, xf[0]);   // (xf[0] was halo row 8's, df[0] output row 6's: both consumed)
          req_d(nbuf, 0, df[0]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int o = 0; o < 3; ++o) {
        const int ky = o == 0 ? 2 : (o == 1 ? 1 : 0);  // ky = 2 first: it frees the ring slot of output row r - 2
        const int y = r - ky;
        if (y >= 0 && y < TH) {
#pragma unroll
          for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int n4 = 0; n4 < 4; ++n4)
              acc[ky * 3 + kx][n4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[r & 1][kx], df[y % 3][n4], acc[ky * 3 + kx][n4], 0, 0, 0);
        }
        if (o == 0) {
          __builtin_amdgcn_sched_barrier(0);
          if (r + 1 < TH) req_d(buf, r + 1, df[(r + 1) % 3]);  // (slot (r + 1) % 3 = (r - 2) % 3: just released)
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  __builtin_amdgcn_s_setprio(0);
  __builtin_amdgcn_s_barrier();  // (the loaders' bias reduction uses the tile buffers: after the last fragment read)

  const int cin = ci0 + cb * 16 + g * 4;
  const int cout = co0 + li;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int n4 = 0; n4 < 4; ++n4) {
      float* dst = out + ((int64_t)tap * Cin + cin) * Cout + cout + n4 * 16;
      if (a.slab) {
        dst[0] = acc[tap][n4].x;
        dst[Cout] = acc[tap][n4].y;
        dst[2 * Cout] = acc[tap][n4].z;
        dst[3 * Cout] = acc[tap][n4].w;
      } else {
        atomicAdd(dst, acc[tap][n4].x);
        atomicAdd(dst + Cout, acc[tap][n4].y);
        atomicAdd(dst + 2 * Cout, acc[tap][n4].z);
        atomicAdd(dst + 3 * Cout, acc[tap][n4].w);
      }
    }
}

// ---- conv1_1 (fp32 input of 1-4 channels, 64 output channels): filter + bias gradient on the bf16 matrix instruction -----
// dW[t][co] = sum_pix X[pix][t] dY[pix][co] with 9 CIN + 1 rows t (row 9 CIN: X = 1, the bias gradient).  The fp32 form
// (backward.hip, v_mfma_f32_16x16x4_f32) is bound by that instruction's rate: 8 of them per 4 pixels.  Here X is split
// EXACTLY into three bf16 terms (hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid): 24 mantissa bits; the forward
// kernel's scheme) and each term meets the bf16 dY on v_mfma_f32_16x16x32_bf16 -- every product exact in fp32, the sums in
// fp32 as before, 3 x 8 instructions of 16 cycles per 32 pixels instead of 64 of 32.  Terms that are zero for the whole wave
// (mid and lo of 8-bit image data) are skipped -- adding exact zeros changes no bit.
// A workgroup walks 8x32-pixel tiles: the dY tile by LDS-DMA into the swizzled [pixel][64 ch] image of the filter-gradient
// kernels above (same transposing fragment reads), the fp32 halo patch of X through registers into LDS; wave w takes image
// row w of the tile as its K-step of 32 pixels and keeps the whole (9 CIN + 1) x 64 block in registers; the eight waves meet
// in LDS in wave order and the workgroup's sums go to its row of `part_out` (fixed-order reduce) or onto dw / db by atomics.
template <int CIN>
__global__ __launch_bounds__(512) void conv_first_wgrad_split_kernel(const float* __restrict__ x, const __bf16* __restrict__ dy,
                                                                    float* __restrict__ dw, float* __restrict__ db, int N, int H,
                                                                    int W, float* __restrict__ part_out) {
  constexpr int TH = 8, TW = 32, HH = TH + 2, HW = TW + 2;
  constexpr int K = 9 * CIN, MB = (K + 1 + 15) / 16;
  constexpr int D_BYTES = TH * TW * 128;
  constexpr int XP = HH * HW * CIN;                      // floats of the X patch
  constexpr int X_BYTES = (XP * 4 + 1023) / 1024 * 1024;
  constexpr int BUF = D_BYTES + X_BYTES;
  constexpr int XS = (XP + 511) / 512;                   // patch elements per thread
  constexpr uint32_t OOB = 0x80000000u;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Wp = W + 2;
  const int tiles_x = W / TW, tiles_y = H / TH, n_tiles = N * tiles_y * tiles_x;

  // dY fragment addresses (as conv_wgrad_dma_kernel): lane group g, element j <-> pixel 16 (g >> 1) + 8 (j >> 2) + 4 (g & 1) + (j & 3)
  const int g = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
  const int xk = 16 * (g >> 1) + 4 * (g & 1) + q;
  int dbase[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) dbase[n] = xk * 128 + (xv_swz(xk, n * 2 + (p >> 1)) << 4) + (p & 1) * 8 + wave * (TW * 128);
  // X fragment: row t = 16 mb + li of the operand -> (dy, dx, ci), the bias row, or nothing; element j of the lane is pixel
  // 16 (g >> 1) + 8 (j >> 2) + 4 (g & 1) + (j & 3) of image row `wave` of the tile
  int xfrag[MB], kind[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int t = 16 * mb + li, tap = t / CIN;
    kind[mb] = t < K ? 0 : (t == K ? 1 : 2);
    const int tdy = tap / 3, tdx = tap % 3, ci = t - tap * CIN;  // (patch coordinates: the halo is already inside)
    xfrag[mb] = D_BYTES + (kind[mb] == 0 ? (((wave + tdy) * HW + (16 * (g >> 1) + 4 * (g & 1) + tdx)) * CIN + ci) * 4 : 0);
  }
  // dY pieces of this wave (1 KB each: pieces wave, wave + 8, ..): per-lane source offsets, computed once
  int dvoff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int idx = (wave + 8 * j) * 64 + lane;
    const int pp = idx >> 3, ps = idx & 7;
    const int py = pp / TW, px = pp - py * TW;
    dvoff[j] = ((py * Wp + px) * 64 + xv_swz(px, ps) * 8) * 2;
  }
  // X patch elements of this thread: offset inside the patch's bounding box of the image and the edge flags that void it
  // (bit 0 first patch row, 1 last, 2 first column, 3 last): the tile's own flags say which of them lie outside the image
  uint32_t xrel[XS], xkill[XS];
#pragma unroll
  for (int e = 0; e < XS; ++e) {
    const int i = tid + 512 * e;
    const int hy = i / (HW * CIN), r = i - hy * (HW * CIN), hx = r / CIN;
    xrel[e] = (uint32_t)((hy * W * CIN + r) * 4);  // bytes behind element (y0 - 1, x0 - 1, 0)
    xkill[e] = i < XP ? ((hy == 0 ? 1u : 0u) | (hy == HH - 1 ? 2u : 0u) | (hx == 0 ? 4u : 0u) | (hx == HW - 1 ? 8u : 0u)) : 16u;
  }
  // descriptor over X starting one row and one pixel BEFORE the tensor (only address arithmetic: the elements there are
  // voided by the edge flags), so every offset is non-negative; the tile's position goes into the scalar offset
  const int back = (W + 1) * CIN * 4;
  const uint64_t xa = (uint64_t)x - back;
  const uint32_t xlo = __builtin_amdgcn_readfirstlane((uint32_t)xa), xhi = __builtin_amdgcn_readfirstlane((uint32_t)(xa >> 32));
  const auto xrs = __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)xhi << 32) | xlo), 0, 0x7fffffff, 0x00020000);

  struct Tile {
    int n, y0, x0;
  };
  auto decode = [&](int t) {
    Tile tl;
    tl.x0 = (t % tiles_x) * TW;
    const int r = t / tiles_x;
    tl.y0 = (r % tiles_y) * TH;
    tl.n = r / tiles_y;
    return tl;
  };
  float xreg[XS];
  auto request = [&](const Tile& tl, int b) {  // dY by LDS-DMA into buffer b, X into registers
    const __bf16* dsrc = dy + (((int64_t)tl.n * (H + 2) + (tl.y0 + 1)) * Wp + (tl.x0 + 1)) * 64;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int lds = __builtin_amdgcn_readfirstlane(b * BUF + (wave + 8 * j) * 1024);
      asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds), "v"(dvoff[j]), "s"(dsrc) : "memory");
    }
    const uint32_t edge = (tl.y0 == 0 ? 1u : 0u) | (tl.y0 + TH == H ? 2u : 0u) | (tl.x0 == 0 ? 4u : 0u) | (tl.x0 + TW == W ? 8u : 0u) | 16u;
    const uint32_t tbase = (uint32_t)(((tl.n * H + tl.y0) * W + tl.x0) * CIN * 4);  // (< 2^31: checked by the launcher)
#pragma unroll
    for (int e = 0; e < XS; ++e)
      xreg[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, (xkill[e] & edge) == 0 ? xrel[e] : OOB, tbase, 0));
  };
  auto publish = [&](int b) {  // the X patch of the requested tile from the registers into buffer b
#pragma unroll
    for (int e = 0; e < XS; ++e)
      if (tid + 512 * e < XP) *reinterpret_cast<float*>(smem + b * BUF + D_BYTES + (tid + 512 * e) * 4) = xreg[e];
  };

  f32x4 acc[MB][4];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[mb][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  int t = blockIdx.x;
  if (t < n_tiles) {
    request(decode(t), 0);
    publish(0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int b = 0;
  for (; t < n_tiles; t += gridDim.x) {
    const bool more = t + (int)gridDim.x < n_tiles;
    if (more) request(decode(t + gridDim.x), b ^ 1);
    const char* buf = smem + b * BUF;
    bf16x8 bfr[4];
#pragma unroll
    for (int n4 = 0; n4 < 4; ++n4) bfr[n4] = tr_read2(buf, dbase[n4], dbase[n4] + 8 * 128);
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const float*>(buf + xfrag[mb] + ((j & 3) + 8 * (j >> 2)) * CIN * 4);
      if (kind[mb] != 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = kind[mb] == 1 ? 1.f : 0.f;
      }
      // x = hi + mid + lo, each a bf16 (round to nearest even; the remainders are exact in fp32)
      uint32_t hi[4], mid[4], lo[4];
      float r1[8], r2[8];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        hi[k] = pack_bf16x2(v[2 * k], v[2 * k + 1]);
        r1[2 * k] = v[2 * k] - __builtin_bit_cast(float, hi[k] << 16);
        r1[2 * k + 1] = v[2 * k + 1] - __builtin_bit_cast(float, hi[k] & 0xffff0000u);
        mid[k] = pack_bf16x2(r1[2 * k], r1[2 * k + 1]);
        r2[2 * k] = r1[2 * k] - __builtin_bit_cast(float, mid[k] << 16);
        r2[2 * k + 1] = r1[2 * k + 1] - __builtin_bit_cast(float, mid[k] & 0xffff0000u);
        lo[k] = pack_bf16x2(r2[2 * k], r2[2 * k + 1]);
      }
      const bf16x8 ah = __builtin_bit_cast(bf16x8, u32x4{hi[0], hi[1], hi[2], hi[3]});
#pragma unroll
      for (int n4 = 0; n4 < 4; ++n4) acc[mb][n4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bfr[n4], acc[mb][n4], 0, 0, 0);
      // (wave-uniform: a term that is zero in every lane -- (mid | lo) & 0x7fff7fff ignores the sign of a zero -- adds nothing)
      const uint32_t any_mid = (mid[0] | mid[1] | mid[2] | mid[3]) & 0x7fff7fffu;
      if (__builtin_amdgcn_ballot_w64(any_mid != 0) != 0) {
        const bf16x8 am = __builtin_bit_cast(bf16x8, u32x4{mid[0], mid[1], mid[2], mid[3]});
#pragma unroll
        for (int n4 = 0; n4 < 4; ++n4) acc[mb][n4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bfr[n4], acc[mb][n4], 0, 0, 0);
        const uint32_t any_lo = (lo[0] | lo[1] | lo[2] | lo[3]) & 0x7fff7fffu;
        if (__builtin_amdgcn_ballot_w64(any_lo != 0) != 0) {
          const bf16x8 al = __builtin_bit_cast(bf16x8, u32x4{lo[0], lo[1], lo[2], lo[3]});
#pragma unroll
          for (int n4 = 0; n4 < 4; ++n4) acc[mb][n4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bfr[n4], acc[mb][n4], 0, 0, 0);
        }
      }
    }
    if (more) publish(b ^ 1);  // (nobody reads buffer b ^ 1 during this tile)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    b ^= 1;
  }
  // accumulator: row 16 mb + 4 g + r, column 16 n4 + li.  The eight waves add their blocks into LDS one after the other
  // (fixed order), the workgroup's sums go to its row of part_out
  float* red = reinterpret_cast<float*>(smem);
  for (int c = tid; c < MB * 16 * 64; c += 512) red[c] = 0.f;
  __syncthreads();
  for (int w = 0; w < 8; ++w) {
    if (wave == w) {
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int n4 = 0; n4 < 4; ++n4)
#pragma unroll
          for (int r = 0; r < 4; ++r) red[(16 * mb + 4 * g + r) * 64 + 16 * n4 + li] += acc[mb][n4][r];
    }
    __syncthreads();
  }
  for (int c = tid; c < (K + 1) * 64; c += 512) {
    const float v = red[c];
    if (part_out != nullptr) {
      part_out[(int64_t)blockIdx.x * (K + 1) * 64 + c] = v;
      continue;
    }
    if (v == 0.f) continue;
    if (c < K * 64)
      atomicAdd(dw + c, v);
    else if (db != nullptr)
      atomicAdd(db + (c - K * 64), v);
  }
}

// ---- the dense 1x1 score layer at full resolution (batch-norm / joint trainers): filter + bias gradient --------------------
// dW[u][c] = sum_pix Y[pix][u] dS[pix][c], db[c] = sum_pix dS[pix][c]: Y the bf16 64-channel feature map, dS the fp32
// class-score gradient ([N][H][W][C] dense, C <= 16).  The scheme of conv_first_wgrad_split_kernel with the roles of a 1x1
// layer: the Y tile (8x32 pixels) by LDS-DMA and transposing fragment reads as the matrix instruction's B operand, dS split
// EXACTLY into three bf16 terms as its A operand (rows = classes), every product exact in fp32; the fp32 form
// (batchnorm.hip, v_mfma_f32_16x16x4_f32: 8 instructions per 4 pixels) ran at 354 us against 150 us of memory traffic.  The
// bias gradient is the lanes' own sum of the dS values they load.  Workgroup sums to part_out[block][(16 + 1) * 64]:
// rows 0..15 = classes x 64 units, row 16 = db in its first 16 entries.
__global__ __launch_bounds__(512) void score_dense_wgrad_split_kernel(const float* __restrict__ ds, const __bf16* __restrict__ y,
                                                                     int N, int H, int W, int C, float* __restrict__ part_out) {
  constexpr int TH = 8, TW = 32;
  constexpr int D_BYTES = TH * TW * 128;
  constexpr int XMAX = TH * TW * 16;  // floats of a dS tile at C = 16
  constexpr int X_BYTES = XMAX * 4;
  constexpr int BUF = D_BYTES + X_BYTES;
  constexpr int XS = XMAX / 512;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Wp = W + 2;
  const int tiles_x = W / TW, tiles_y = H / TH, n_tiles = N * tiles_y * tiles_x;
  const int XP = TH * TW * C;
  const int g = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
  const int xk = 16 * (g >> 1) + 4 * (g & 1) + q;
  int dbase[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) dbase[n] = xk * 128 + (xv_swz(xk, n * 2 + (p >> 1)) << 4) + (p & 1) * 8 + wave * (TW * 128);
  // dS fragment: row li = class; element j of the lane = pixel 16 (g >> 1) + 8 (j >> 2) + 4 (g & 1) + (j & 3) of image row `wave`
  const bool live = li < C;
  const int xfrag = D_BYTES + (live ? ((wave * TW + 16 * (g >> 1) + 4 * (g & 1)) * C + li) * 4 : 0);
  int dvoff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int idx = (wave + 8 * j) * 64 + lane;
    const int pp = idx >> 3, ps = idx & 7;
    const int py = pp / TW, px = pp - py * TW;
    dvoff[j] = ((py * Wp + px) * 64 + xv_swz(px, ps) * 8) * 2;
  }
  // dS tile elements of this thread: row r of the tile (32 pixels x C floats, contiguous in memory), offset inside the row
  uint32_t xrel[XS];
#pragma unroll
  for (int e = 0; e < XS; ++e) {
    const int i = tid + 512 * e;
    const int row = i / (TW * C), r = i - row * (TW * C);
    xrel[e] = i < XP ? (uint32_t)((row * W * C + r) * 4) : 0x80000000u;
  }
  const uint64_t xa = (uint64_t)ds;
  const uint32_t xlo = __builtin_amdgcn_readfirstlane((uint32_t)xa), xhi = __builtin_amdgcn_readfirstlane((uint32_t)(xa >> 32));
  const auto xrs = __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)xhi << 32) | xlo), 0, 0x7fffffff, 0x00020000);
  struct Tile {
    int n, y0, x0;
  };
  auto decode = [&](int t) {
    Tile tl;
    tl.x0 = (t % tiles_x) * TW;
    const int r = t / tiles_x;
    tl.y0 = (r % tiles_y) * TH;
    tl.n = r / tiles_y;
    return tl;
  };
  float xreg[XS];
  auto request = [&](const Tile& tl, int b) {
    const __bf16* dsrc = y + (((int64_t)tl.n * (H + 2) + (tl.y0 + 1)) * Wp + (tl.x0 + 1)) * 64;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int lds = __builtin_amdgcn_readfirstlane(b * BUF + (wave + 8 * j) * 1024);
      asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds), "v"(dvoff[j]), "s"(dsrc) : "memory");
    }
    const uint32_t tbase = (uint32_t)(((tl.n * H + tl.y0) * W + tl.x0) * C * 4);  // (< 2^31: checked by the launcher)
#pragma unroll
    for (int e = 0; e < XS; ++e) xreg[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, xrel[e], tbase, 0));
  };
  auto publish = [&](int b) {
#pragma unroll
    for (int e = 0; e < XS; ++e)
      if (tid + 512 * e < XP) *reinterpret_cast<float*>(smem + b * BUF + D_BYTES + (tid + 512 * e) * 4) = xreg[e];
  };
  f32x4 acc[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;
  int t = blockIdx.x;
  if (t < n_tiles) {
    request(decode(t), 0);
    publish(0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int b = 0;
  for (; t < n_tiles; t += gridDim.x) {
    const bool more = t + (int)gridDim.x < n_tiles;
    if (more) request(decode(t + gridDim.x), b ^ 1);
    const char* buf = smem + b * BUF;
    bf16x8 bfr[4];
#pragma unroll
    for (int n4 = 0; n4 < 4; ++n4) bfr[n4] = tr_read2(buf, dbase[n4], dbase[n4] + 8 * 128);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = live ? *reinterpret_cast<const float*>(buf + xfrag + ((j & 3) + 8 * (j >> 2)) * C * 4) : 0.f;
    bsum += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    uint32_t hi[4], mid[4], lo[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      hi[k] = pack_bf16x2(v[2 * k], v[2 * k + 1]);
      const float r0 = v[2 * k] - __builtin_bit_cast(float, hi[k] << 16), r1 = v[2 * k + 1] - __builtin_bit_cast(float, hi[k] & 0xffff0000u);
      mid[k] = pack_bf16x2(r0, r1);
      lo[k] = pack_bf16x2(r0 - __builtin_bit_cast(float, mid[k] << 16), r1 - __builtin_bit_cast(float, mid[k] & 0xffff0000u));
    }
    const bf16x8 ah = __builtin_bit_cast(bf16x8, u32x4{hi[0], hi[1], hi[2], hi[3]});
    const bf16x8 am = __builtin_bit_cast(bf16x8, u32x4{mid[0], mid[1], mid[2], mid[3]});
    const bf16x8 al = __builtin_bit_cast(bf16x8, u32x4{lo[0], lo[1], lo[2], lo[3]});
#pragma unroll
    for (int n4 = 0; n4 < 4; ++n4) acc[n4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bfr[n4], acc[n4], 0, 0, 0);
#pragma unroll
    for (int n4 = 0; n4 < 4; ++n4) acc[n4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bfr[n4], acc[n4], 0, 0, 0);
#pragma unroll
    for (int n4 = 0; n4 < 4; ++n4) acc[n4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bfr[n4], acc[n4], 0, 0, 0);
    if (more) publish(b ^ 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    b ^= 1;
  }
  // accumulator: row (class) 4 g + r, column (unit) 16 n4 + li; bsum: class li, one of 4 x 8 partial sums (g, wave)
  float* red = reinterpret_cast<float*>(smem);
  for (int c = tid; c < 17 * 64; c += 512) red[c] = 0.f;
  __syncthreads();
  for (int w = 0; w < 8; ++w) {
    if (wave == w) {
#pragma unroll
      for (int n4 = 0; n4 < 4; ++n4)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(4 * g + r) * 64 + 16 * n4 + li] += acc[n4][r];
      for (int gg = 0; gg < 4; ++gg) {  // the four pixel groups of a class, in order
        if (g == gg) red[16 * 64 + li] += bsum;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
    }
    __syncthreads();
  }
  for (int c = tid; c < 17 * 64; c += 512) part_out[(int64_t)blockIdx.x * (17 * 64) + c] = red[c];
}

// ---- 1x1 filter gradient as a flat GEMM (round 6): dW[cin][cout] = sum over ALL padded pixels m of X[m][cin] dY[m][cout] ----
// AdapNet's block stages are 1x1 convs (and its atrous pairs in training: a 1x1 conv over the 18C im2col operand, cin up to
// 9 216): conv_wgrad_kernel<1> above gave them 64 x 64 blocks of dW per workgroup -- X re-read cout/64 times and dY cin/64
// times through registers, 32 FLOP per byte -- 24 % of the AdapNet training step.  Here the padded maps are what they are in
// memory, two flat matrices [Mp][C] whose border rows are zero in X, so the reduction runs over consecutive rows with no
// tile geometry: a workgroup owns 256 cin x 128 cout (or 128 x 256) of dW and a slice of the rows (split-K into slabs, added in a fixed
// order by slab_reduce_kernel), eight waves of 64 x 64, 64 rows per step as six [64 rows][64 channels] LDS images (four of
// X, two of dY) filled by LDS-DMA three stages deep behind a counted vmcnt -- the wide flat GEMM's pipeline
// (conv1x1_gemm.hip) with the transposing fragment reads of the kernels above.
constexpr int W1_STAGE = 6 * 8192, W1_STAGES = 3, W1_LDS = W1_STAGES * W1_STAGE;

struct Wgrad1Args {
  const __bf16* x;
  const __bf16* dy;
  float* slab;   // [splits][Cin][Cout]
  float* bslab;  // [splits][Cout] or null
  int64_t Mp;
  int Cin, Cout, n_co, splits, steps_per_split;
  int H, W, d1, d2;  // PAIR: map size and the two dilation rates
};

// GI = 4: 256 cin x 128 cout per workgroup (four X images, two dY images per stage); GI = 2: 128 cin x 256 cout.
// PAIR (GI = 4): the filter gradients of AdapNet's two atrous 3x3 convs (adapnet.py:84-88) WITHOUT the 18C im2col operand: the
// "cin" axis is virtual, (tap, channel) of one conv, and an X image of a stage -- 64 channels of ONE tap -- is gathered by
// the DMA from the map itself, row m reading padded pixel m + off(tap) (or pixel 0, a zero border pixel, where the tap leaves
// the image).  Only the two diagonal blocks of the [18C][F] product exist: a workgroup's four tap-channel groups and its 128
// output channels belong to the same conv; its slab is that conv's own [9C][F/2] matrix = the HWIO gradient of its kernel.
// a.Cin = C (channels of the map), a.Cout = F (channels of dY, both halves); slabs [conv][split][9C][F/2].
template <int GI, bool PAIR = false>
__global__ __launch_bounds__(512, 2) void conv_wgrad_1x1_gemm_kernel(Wgrad1Args a) {
  static_assert(!PAIR || GI == 4, "the pair form is the 256 x 128 layout");
  constexpr int GO = 6 - GI;
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = GI == 4 ? wave >> 1 : wave >> 2, wc = GI == 4 ? wave & 1 : wave & 3;
  const int Cin = a.Cin, Cout = a.Cout;
  const int split = blockIdx.x % a.splits;
  int pair = blockIdx.x / a.splits;
  // PAIR: n_co = 128-channel blocks of ONE half; tiles of one conv first, then the other's
  const int half = Cout >> 1, vcin = 9 * Cin;                       // (PAIR) channels of one conv's output, its virtual cin
  const int tiles_half = PAIR ? (vcin >> 8) * a.n_co : 0;
  const int conv = PAIR && pair >= tiles_half ? 1 : 0;
  if (PAIR) pair -= conv * tiles_half;
  const int co0 = (pair % a.n_co) * (64 * GO), ci0 = (pair / a.n_co) * (64 * GI);
  const int64_t m_begin = (int64_t)split * a.steps_per_split * 64;
  int64_t m_end = m_begin + (int64_t)a.steps_per_split * 64;
  m_end = m_end < a.Mp ? m_end : a.Mp;
  const int nsteps = m_begin < m_end ? (int)((m_end - m_begin + 63) >> 6) : 0;

  // DMA: wave w moves rows 8w .. 8w+7 of each of the six images; lane -> (row, physical 16-byte slot), logical slot through
  // the swizzle of the transposing reads (xv_swz: slot ^ (row & 6))
  const int drow = wave * 8 + (lane >> 3), dslot = (lane & 7) ^ (drow & 6);
  int voff[6];
#pragma unroll
  for (int k = 0; k < 6; ++k)
    voff[k] = (k < GI ? (PAIR ? 0 : ci0 + k * 64) : (conv * half + co0 + (k - GI) * 64)) * 2 + dslot * 16;
  // PAIR: image k is 64 channels of tap tap_k: displacement in padded pixels, channel block
  int tdy[4], tdx[4];
  if (PAIR) {
    const int cpt = Cin >> 6, dil = conv ? a.d2 : a.d1;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int gidx = (ci0 >> 6) + k, tap = gidx / cpt, cb = gidx - tap * cpt, ky = tap / 3, kx = tap - 3 * ky;
      tdy[k] = (ky - 1) * dil, tdx[k] = (kx - 1) * dil;
      voff[k] += cb * 128;
    }
  }
  auto issue = [&](int step, int stage) {
    const int64_t m = m_begin + (int64_t)step * 64;
    // rows past the end of the maps: the last padded pixel, a border pixel (zero in X: no contribution)
    const int64_t rowi = m + drow < a.Mp ? m + drow : a.Mp - 1;
    const int rx = (int)(rowi - m) * Cin * 2, rd = (int)(rowi - m) * Cout * 2;
    const char* xs = reinterpret_cast<const char*>(a.x) + (PAIR ? 0 : m * Cin * 2);
    const char* ds = reinterpret_cast<const char*>(a.dy) + m * Cout * 2;
    const int dst = stage * W1_STAGE + wave * 1024;
    if (PAIR) {
      // the row's padded coordinates; a row that is itself a border pixel (its dY is zero) or past the end reads pixel 0
      const int Wp = a.W + 2, Hp = a.H + 2;
      const int mi = (int)rowi, prow = mi / Wp, xx = mi - prow * Wp, yy = prow % Hp;
      const bool interior = m + drow < a.Mp && xx >= 1 && xx <= a.W && yy >= 1 && yy <= a.H;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int sy = yy + tdy[k], sx = xx + tdx[k];
        const bool ok = interior && sy >= 1 && sy <= a.H && sx >= 1 && sx <= a.W;
        const int src = ok ? (mi + tdy[k] * Wp + tdx[k]) * Cin * 2 : 0;
        asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst + k * 8192), "v"(voff[k] + src), "s"(xs)
                     : "memory");
      }
    } else {
#pragma unroll
    for (int k = 0; k < GI; ++k)
      asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst + k * 8192), "v"(voff[k] + rx), "s"(xs)
                   : "memory");
    }
#pragma unroll
    for (int k = GI; k < 6; ++k)
      asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst + k * 8192), "v"(voff[k] + rd), "s"(ds)
                   : "memory");
  };

  // fragments: lane roles of a transposing read as in the kernels above (pixel xk of a 32-row half, 4-channel piece p)
  const int g = lane >> 4, li = lane & 15, q = li >> 2, pc = li & 3;
  const int xk = 16 * (g >> 1) + 4 * (g & 1) + q;
  int fo[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) fo[i] = xk * 128 + (xv_swz(xk, i * 2 + (pc >> 1)) << 4) + (pc & 1) * 8;
  const int xfrag = wr * 8192, dfrag = (GI + wc) * 8192;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // BiasAddGrad rides along in the workgroups of cin block 0: thread -> (cout = tid % (64 GO), row part = tid / (64 GO))
  const bool do_bias = !PAIR && a.bslab != nullptr && ci0 == 0;
  float bsum = 0.f;

  if (nsteps > 0) issue(0, 0);
  if (nsteps > 1) issue(1, 1);
  int stage = 0;
  for (int step = 0; step < nsteps; ++step) {
    if (step + 1 < nsteps)
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int nstage = stage == 0 ? 2 : stage - 1;  // (step + 2) % 3
    if (step + 2 < nsteps) issue(step + 2, nstage);
    const char* sb = smem + stage * W1_STAGE;
    if (do_bias) {
      constexpr int NCO = 64 * GO, PARTS = 512 / NCO, RPP = 64 / PARTS;   // 128 couts x 4 parts of 16 rows / 256 x 2 of 32
      const int co = tid % NCO, qt = tid / NCO;
      const char* dimg = sb + (GI + (co >> 6)) * 8192 + (co & 7) * 2;
      const int slot = (co & 63) >> 3;
#pragma unroll 4
      for (int r = qt * RPP; r < qt * RPP + RPP; ++r)
        bsum += (float)*reinterpret_cast<const __bf16*>(dimg + r * 128 + (xv_swz(r, slot) << 4));
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      bf16x8 af[4], bf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = tr_read2(sb, xfrag + h * 4096 + fo[i], xfrag + h * 4096 + fo[i] + 8 * 128);
#pragma unroll
      for (int j = 0; j < 4; ++j) bf[j] = tr_read2(sb, dfrag + h * 4096 + fo[j], dfrag + h * 4096 + fo[j] + 8 * 128);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    stage = stage == 2 ? 0 : stage + 1;
  }

  if (do_bias) {
    __syncthreads();  // (the last step's images are consumed)
    float* red = reinterpret_cast<float*>(smem);
    red[tid] = bsum;
    __syncthreads();
    if (GI == 4) {
      if (tid < 128) a.bslab[(int64_t)split * Cout + co0 + tid] = ((red[tid] + red[128 + tid]) + red[256 + tid]) + red[384 + tid];
    } else {
      if (tid < 256) a.bslab[(int64_t)split * Cout + co0 + tid] = red[tid] + red[256 + tid];
    }
  }
  // accumulator (row = cin = 4 (lane >> 4) + r, column = cout = lane & 15) -> slab [cin][cout]
  const int ldo = PAIR ? half : Cout;   // row pitch of the slab
  float* const out = PAIR ? a.slab + ((int64_t)conv * a.splits + split) * vcin * half : a.slab + (int64_t)split * Cin * Cout;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float* dst = out + (int64_t)(ci0 + wr * 64 + i * 16 + g * 4) * ldo + co0 + wc * 64 + j * 16 + li;
      dst[0] = acc[i][j].x;
      dst[ldo] = acc[i][j].y;
      dst[2 * ldo] = acc[i][j].z;
      dst[3 * ldo] = acc[i][j].w;
    }
}

// Split count of the kernel above: the one that needs the fewest (rounds of workgroups) x (steps per workgroup), one
// workgroup per CU.  A pure function of the shape: the workspace query and the launcher must agree.
static int wgrad1_gi(int cin, int cout) { return ((cin & 255) == 0 && (cout & 127) == 0) ? 4 : (((cin & 127) == 0 && (cout & 255) == 0) ? 2 : 0); }
static int wgrad1_splits_of(int64_t mp, int64_t tiles, int* steps_per_split);
static int wgrad1_splits(int64_t mp, int cin, int cout, int* steps_per_split) {
  const int gi = wgrad1_gi(cin, cout);
  return wgrad1_splits_of(mp, (int64_t)(cin / (64 * gi)) * (cout / (64 * (6 - gi))), steps_per_split);
}
static int wgrad1_splits_of(int64_t mp, int64_t tiles, int* steps_per_split) {
  const int64_t steps = (mp + 63) / 64;
  const int cus = xv_num_cus();
  int best = 1;
  int64_t best_cost = -1;
  for (int sp = 1; sp <= 64 && sp <= steps; ++sp) {
    const int64_t per = (steps + sp - 1) / sp, rounds = (tiles * sp + cus - 1) / cus;
    const int64_t cost = rounds * (per + 3);  // (+3: pipeline fill and the slab's stores)
    if (best_cost < 0 || cost < best_cost) best = sp, best_cost = cost;
  }
  if (steps_per_split) *steps_per_split = (int)((steps + best - 1) / best);
  return best;
}
static bool wgrad1_ok(int64_t mp, int cin, int cout) {
  static const bool off = getenv("XV_WGRAD_1X1_GEMM") != nullptr && atoi(getenv("XV_WGRAD_1X1_GEMM")) == 0;  // A/B switch
  return !off && wgrad1_gi(cin, cout) != 0 && mp >= 64 && 64LL * (cin > cout ? cin : cout) * 2 < 0x7fffffffLL;
}

// dw[i] += sum_s slab[s][i], splits summed in a fixed order (bitwise reproducible filter gradients)
// (and db[c] += sum_s bslab[s][c], the bias gradient's partial sums, in the same launch).  SUBS lanes share an element:
// lane `sub` adds splits sub, sub + SUBS, ... in order, then a butterfly over the SUBS partial sums -- a fixed tree either
// way.  With many splits of a small kernel (the 64-channel and 1x1 layers: up to 256 splits of 36 K elements) one thread
// per element is a chain of `splits` dependent memory round trips on a handful of workgroups: 16 lanes per element cut
// it 16-fold.
template <int SUBS>
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw,
                                                         int64_t n4, int splits, const float* __restrict__ bslab,
                                                         float* __restrict__ db, int cout4) {
  const int sub = threadIdx.x % SUBS;
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) / SUBS; i < n4 + cout4; i += (int64_t)gridDim.x * 256 / SUBS) {
    const bool bias = i >= n4;
    const int64_t j = bias ? i - n4 : i, stride = bias ? cout4 : n4;
    const float* src = bias ? bslab : slab;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int k = sub; k < splits; k += SUBS) s += *reinterpret_cast<const f32x4*>(src + ((int64_t)k * stride + j) * 4);
#pragma unroll
    for (int off = SUBS / 2; off >= 1; off >>= 1) {
      s.x += __shfl_xor(s.x, off, 64);
      s.y += __shfl_xor(s.y, off, 64);
      s.z += __shfl_xor(s.z, off, 64);
      s.w += __shfl_xor(s.w, off, 64);
    }
    if (sub == 0) {
      f32x4* d = reinterpret_cast<f32x4*>((bias ? db : dw) + j * 4);
      *d = *d + s;
    }
  }
}

// db[c] += sum over all pixels of dY[.., c]   (bias gradient; dY border is zero so the padded
// buffer is summed as a flat [rows][C] matrix)
__global__ __launch_bounds__(256) void bias_grad_kernel(const __bf16* __restrict__ dy, int64_t rows, int C,
                                                       float* __restrict__ db) {
  // thread -> 8-channel group cg = tid % (C/8), row lane = tid / (C/8)
  const int c8 = C >> 3;
  const int cg = threadIdx.x % c8;
  const int rl = threadIdx.x / c8;
  const int rstep = 256 / c8;
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int64_t r = (int64_t)blockIdx.x * rstep + rl; r < rows; r += (int64_t)gridDim.x * rstep) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(dy + r * C + cg * 8);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      s[2 * i] += bf16_bits_to_f32(v[i] & 0xffffu);
      s[2 * i + 1] += __builtin_bit_cast(float, v[i] & 0xffff0000u);
    }
  }
  __shared__ float red[256 * 8];
#pragma unroll
  for (int i = 0; i < 8; ++i) red[threadIdx.x * 8 + i] = s[i];
  __syncthreads();
  if (rl == 0) {
    for (int o = 1; o < rstep; ++o)
#pragma unroll
      for (int i = 0; i < 8; ++i) s[i] += red[(o * c8 + cg) * 8 + i];
#pragma unroll
    for (int i = 0; i < 8; ++i) atomicAdd(db + cg * 8 + i, s[i]);
  }
}

}  // namespace

extern "C" int xv_bias_grad(const xv_act* dy, float* dbias, void* stream) {
  XV_REQUIRE_BF16(dy);
  XV_CHECK_ARG(dy && dy->data && dbias);
  XV_CHECK_SHAPE(dy->c > 0 && (dy->c & 7) == 0 && dy->c <= 2048 && 256 % (dy->c >> 3) == 0);
  const int64_t rows = (int64_t)dy->n * (dy->h + 2) * (dy->w + 2);
  int64_t blocks = (rows + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(bias_grad_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     (const __bf16*)dy->data, rows, dy->c, dbias);
  return xv_launch_status();
}

static int wgrad_default_variant() {
  const char* e = getenv("XV_WGRAD_VARIANT");  // A/B timing: 1, 2 (round 5's default) or 3
  const int v = e != nullptr ? atoi(e) : 3;
  return v >= 1 && v <= 3 ? v : 3;
}
static int g_wgrad_variant = wgrad_default_variant();
extern "C" int xv_set_wgrad_variant(int v) {
  if (v == 0) v = wgrad_default_variant();
  if (v < 1 || v > 3) return XV_EINVAL;
  g_wgrad_variant = v;
  return XV_OK;
}

static int wgrad_splits(int k, int pairs, int n_ptiles) {
  // 3x3, LDS-DMA kernel: one 8-wave workgroup per CU; otherwise ~3 four-wave workgroups per CU
  int sp = (g_wgrad_variant >= 2 && k == 3) ? (xv_num_cus() + pairs - 1) / pairs : (3 * xv_num_cus() + pairs - 1) / pairs;
  if (sp > n_ptiles) sp = n_ptiles;
  return sp < 1 ? 1 : sp;
}

extern "C" size_t xv_conv2d_bwd_filter_workspace_bytes(int n, int h, int w, int cin, int cout, int k) {
  if (!xv_dims_sane(n, h, w) || cin <= 0 || cout <= 0 || (cin & 63) || (cout & 63) || (k != 1 && k != 3)) return 0;
  const int64_t ptiles = (((int64_t)w + 31) / 32) * (((int64_t)h + 7) / 8) * n;
  const int64_t pairs = (int64_t)(cin >> 6) * (cout >> 6);
  // upper bound over both kernel variants
  int64_t sp = (3 * (int64_t)xv_num_cus() + pairs - 1) / pairs;
  if (sp > ptiles) sp = ptiles;
  if (sp < 1) sp = 1;
  if (k == 1 && wgrad1_ok((int64_t)n * (h + 2) * (w + 2), cin, cout)) {
    const int64_t sp1 = wgrad1_splits((int64_t)n * (h + 2) * (w + 2), cin, cout, nullptr);
    if (sp1 > sp) sp = sp1;
  }
  return (size_t)sp * ((size_t)k * k * cin * cout + cout) * sizeof(float);  // dW slabs + bias slabs
}

// ---- filter gradients of AdapNet's atrous pair without the im2col operand (conv_wgrad_1x1_gemm_kernel<4, true>) ----------
static bool wgrad_pair_ok(int64_t mp, int c, int f) {
  return (c & 255) == 0 && (f & 255) == 0 && mp >= 64 && mp * (c > f ? c : f) * 2 < 0x7fffffffLL;
}
static int64_t wgrad_pair_tiles(int c, int f) { return 2LL * (9 * c / 256) * (f / 256); }

extern "C" size_t xv_conv_dilated_pair_bwd_filter_workspace_bytes(int n, int h, int w, int c, int f) {
  if (!xv_dims_sane(n, h, w) || c <= 0 || f <= 0) return 0;
  const int64_t mp = (int64_t)n * (h + 2) * (w + 2);
  if (!wgrad_pair_ok(mp, c, f)) return 0;
  return (size_t)wgrad1_splits_of(mp, wgrad_pair_tiles(c, f), nullptr) * 2 * 9 * c * (f / 2) * sizeof(float);
}

// dw1 / dw2 (+=): HWIO [3][3][C][F/2] gradients of the kernels of the two convs (rates dilation1 / dilation2) whose outputs are
// the two channel halves of dy; x [N,H,W,C].  Split-K slabs in the workspace, added in a fixed order: bitwise reproducible.
extern "C" int xv_conv_dilated_pair_bwd_filter_ws(const xv_act* x, const xv_act* dy, int dilation1, int dilation2, float* dw1,
                                                  float* dw2, void* workspace, size_t workspace_bytes, void* stream) {
  XV_REQUIRE_BF16(x, dy);
  XV_CHECK_ARG(x && dy && x->data && dy->data && dw1 && dw2 && workspace);
  XV_CHECK_SHAPE(dy->n == x->n && dy->h == x->h && dy->w == x->w && x->h > 0 && dilation1 >= 1 && dilation2 >= 1);
  const int64_t mp = (int64_t)x->n * (x->h + 2) * (x->w + 2);
  XV_CHECK_SHAPE(wgrad_pair_ok(mp, x->c, dy->c));
  XV_CHECK_ARG((((uintptr_t)x->data | (uintptr_t)dy->data | (uintptr_t)dw1 | (uintptr_t)dw2 | (uintptr_t)workspace) & 15) == 0);
  Wgrad1Args g{};
  g.x = (const __bf16*)x->data, g.dy = (const __bf16*)dy->data, g.Mp = mp, g.Cin = x->c, g.Cout = dy->c;
  g.n_co = dy->c / 256;  // 128-channel blocks of one half
  g.H = x->h, g.W = x->w, g.d1 = dilation1, g.d2 = dilation2;
  const int64_t tiles = wgrad_pair_tiles(x->c, dy->c);
  g.splits = wgrad1_splits_of(mp, tiles, &g.steps_per_split);
  const int64_t per_conv = 9LL * x->c * (dy->c / 2);
  if (workspace_bytes < (size_t)g.splits * 2 * per_conv * sizeof(float)) return XV_EWORKSPACE;
  g.slab = (float*)workspace, g.bslab = nullptr;
  hipStream_t s = (hipStream_t)stream;
  static bool attrp[XV_MAX_DEVICES] = {false};
  const hipError_t e = xv_allow_dynamic_lds(reinterpret_cast<const void*>(&conv_wgrad_1x1_gemm_kernel<4, true>), W1_LDS, attrp);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL((conv_wgrad_1x1_gemm_kernel<4, true>), dim3((unsigned)(tiles * g.splits)), dim3(512), W1_LDS, s, g);
  int rc = xv_launch_status();
  if (rc != XV_OK) return rc;
  const int64_t n4 = per_conv / 4;
  for (int conv = 0; conv < 2; ++conv) {
    const float* src = g.slab + (int64_t)conv * g.splits * per_conv;
    float* dst = conv ? dw2 : dw1;
    if (g.splits >= 16) {
      int64_t blocks = (n4 * 16 + 255) / 256;
      if (blocks > 4096) blocks = 4096;
      hipLaunchKernelGGL(slab_reduce_kernel<16>, dim3((unsigned)blocks), dim3(256), 0, s, src, dst, n4, g.splits, (const float*)nullptr,
                         (float*)nullptr, 0);
    } else {
      int64_t blocks = (n4 + 255) / 256;
      if (blocks > 2048) blocks = 2048;
      hipLaunchKernelGGL(slab_reduce_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, s, src, dst, n4, g.splits, (const float*)nullptr,
                         (float*)nullptr, 0);
    }
  }
  return xv_launch_status();
}

extern "C" int xv_conv2d_bwd_filter_ws(const xv_act* x, const xv_act* dy, float* dw_hwio, float* dbias, int k,
                                       void* workspace, size_t workspace_bytes, void* stream);

extern "C" int xv_conv2d_bwd_filter(const xv_act* x, const xv_act* dy, float* dw_hwio, float* dbias, int k,
                                    void* stream) {
  XV_REQUIRE_BF16(x, dy);
  return xv_conv2d_bwd_filter_ws(x, dy, dw_hwio, dbias, k, nullptr, 0, stream);
}

extern "C" int xv_conv2d_bwd_filter_ws(const xv_act* x, const xv_act* dy, float* dw_hwio, float* dbias, int k,
                                       void* workspace, size_t workspace_bytes, void* stream) {
  XV_REQUIRE_BF16(x, dy);
  XV_CHECK_ARG(x && dy && x->data && dy->data && dw_hwio);
  XV_CHECK_SHAPE(k == 1 || k == 3);
  XV_CHECK_SHAPE(x->n > 0 && x->h > 0 && x->w > 0 && x->c > 0 && (x->c & 63) == 0 && (dy->c & 63) == 0 && dy->c > 0);
  XV_CHECK_SHAPE(dy->n == x->n && dy->h == x->h && dy->w == x->w);
  XV_CHECK_ARG((((uintptr_t)x->data | (uintptr_t)dy->data | (uintptr_t)dw_hwio) & 15) == 0);
  WgradArgs a{};
  a.x = (const __bf16*)x->data;
  a.dy = (const __bf16*)dy->data;
  a.dw = dw_hwio;
  a.db = dbias;
  a.N = x->n;
  a.H = x->h;
  a.W = x->w;
  a.Cin = x->c;
  a.Cout = dy->c;
  a.tiles_x = (a.W + 31) / 32;
  a.tiles_y = (a.H + 7) / 8;
  const int64_t ptiles = (int64_t)a.tiles_x * a.tiles_y * a.N;
  XV_CHECK_SHAPE(ptiles <= 0x7fffffff);
  a.n_ptiles = (int)ptiles;
  hipStream_t s = (hipStream_t)stream;
  const int64_t mp = (int64_t)a.N * (a.H + 2) * (a.W + 2);
  if (k == 1 && workspace != nullptr && wgrad1_ok(mp, a.Cin, a.Cout)) {
    // the flat-GEMM form (conv_wgrad_1x1_gemm_kernel): slabs always, added in a fixed order
    Wgrad1Args g{};
    const int gi = wgrad1_gi(a.Cin, a.Cout);
    g.x = a.x, g.dy = a.dy, g.Mp = mp, g.Cin = a.Cin, g.Cout = a.Cout, g.n_co = a.Cout / (64 * (6 - gi));
    g.splits = wgrad1_splits(mp, a.Cin, a.Cout, &g.steps_per_split);
    const int64_t dw1 = (int64_t)a.Cin * a.Cout;
    if (workspace_bytes < (size_t)g.splits * (dw1 + a.Cout) * sizeof(float)) return XV_EWORKSPACE;
    XV_CHECK_ARG(((uintptr_t)workspace & 15) == 0);
    g.slab = (float*)workspace;
    g.bslab = dbias != nullptr ? g.slab + (size_t)g.splits * dw1 : nullptr;
    static bool attr4[XV_MAX_DEVICES] = {false}, attr2[XV_MAX_DEVICES] = {false};
    const unsigned grid1 = (unsigned)((a.Cin / (64 * gi)) * g.n_co * g.splits);
    if (gi == 4) {
      const hipError_t e = xv_allow_dynamic_lds(reinterpret_cast<const void*>(&conv_wgrad_1x1_gemm_kernel<4>), W1_LDS, attr4);
      if (e != hipSuccess) return (int)e;
      hipLaunchKernelGGL(conv_wgrad_1x1_gemm_kernel<4>, dim3(grid1), dim3(512), W1_LDS, s, g);
    } else {
      const hipError_t e = xv_allow_dynamic_lds(reinterpret_cast<const void*>(&conv_wgrad_1x1_gemm_kernel<2>), W1_LDS, attr2);
      if (e != hipSuccess) return (int)e;
      hipLaunchKernelGGL(conv_wgrad_1x1_gemm_kernel<2>, dim3(grid1), dim3(512), W1_LDS, s, g);
    }
    int rc = xv_launch_status();
    if (rc != XV_OK) return rc;
    const int64_t n4 = dw1 / 4;
    const int cout4 = g.bslab != nullptr ? a.Cout / 4 : 0;
    int64_t blocks = (n4 + cout4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (g.splits >= 16) {
      blocks = ((n4 + cout4) * 16 + 255) / 256;
      if (blocks > 4096) blocks = 4096;
      hipLaunchKernelGGL(slab_reduce_kernel<16>, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)g.slab, dw_hwio, n4, g.splits,
                         (const float*)g.bslab, dbias, cout4);
    } else {
      hipLaunchKernelGGL(slab_reduce_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)g.slab, dw_hwio, n4, g.splits,
                         (const float*)g.bslab, dbias, cout4);
    }
    return xv_launch_status();
  }
  const int pairs = (a.Cin >> 6) * (a.Cout >> 6);
  const int splits = wgrad_splits(k, pairs, a.n_ptiles);
  a.splits = splits;
  const int64_t dw_elems = (int64_t)k * k * a.Cin * a.Cout;
  a.slab = a.bslab = nullptr;
  // With a workspace EVERY partial sum -- the dW blocks of the pixel splits and the bias gradient's -- goes to slabs that a
  // second kernel adds in a fixed order: bitwise reproducible gradients.  (Round 2 kept fp32 atomics for the layers with
  // 1-2 block pairs and for the 1x1 layers, where they are a few per cent faster: XV_WGRAD_ATOMICS=1 restores that for
  // A/B timing.)  Without a workspace: atomics.
  static const bool atomics_ok = getenv("XV_WGRAD_ATOMICS") != nullptr;
  if (workspace != nullptr && !(atomics_ok && !(splits > 1 && k == 3 && pairs >= 4))) {
    if (workspace_bytes < (size_t)splits * (dw_elems + a.Cout) * sizeof(float)) return XV_EWORKSPACE;
    XV_CHECK_ARG(((uintptr_t)workspace & 15) == 0);
    a.slab = (float*)workspace;
    a.bslab = dbias != nullptr ? a.slab + (size_t)splits * dw_elems : nullptr;
  }
  auto finish = [&]() -> int {
    int rc = xv_launch_status();
    if (rc != XV_OK || a.slab == nullptr) return rc;
    const int64_t n4 = dw_elems / 4;
    const int cout4 = a.bslab != nullptr ? a.Cout / 4 : 0;
    if (splits >= 16) {
      int64_t blocks = ((n4 + cout4) * 16 + 255) / 256;
      if (blocks > 4096) blocks = 4096;
      hipLaunchKernelGGL(slab_reduce_kernel<16>, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)a.slab, dw_hwio, n4,
                         splits, (const float*)a.bslab, dbias, cout4);
    } else {
      int64_t blocks = (n4 + cout4 + 255) / 256;
      if (blocks > 2048) blocks = 2048;
      hipLaunchKernelGGL(slab_reduce_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)a.slab, dw_hwio, n4,
                         splits, (const float*)a.bslab, dbias, cout4);
    }
    return xv_launch_status();
  };
  if (g_wgrad_variant == 3 && k == 3) {
    constexpr int lds = 2 * (((10 * 34 * 128 + 1023) / 1024) * 1024 + 8 * 32 * 128);
    static bool attr[XV_MAX_DEVICES] = {false};
    {
      const hipError_t e = xv_allow_dynamic_lds(reinterpret_cast<const void*>(&conv_wgrad_lw_kernel), lds, attr);
      if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(conv_wgrad_lw_kernel, dim3((unsigned)(pairs * splits)), dim3(512), lds, s, a);
    return finish();
  }
  if (g_wgrad_variant == 2 && k == 3) {
    constexpr int lds = 2 * (((10 * 34 * 128 + 1023) / 1024) * 1024 + 8 * 32 * 128);
    static bool attr[XV_MAX_DEVICES] = {false};
    {
      const hipError_t e = xv_allow_dynamic_lds(reinterpret_cast<const void*>(&conv_wgrad_dma_kernel<3>), lds, attr);
      if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(conv_wgrad_dma_kernel<3>, dim3((unsigned)(pairs * splits)), dim3(512), lds, s, a);
    return finish();
  }
  const unsigned grid = (unsigned)(pairs * splits);
  if (k == 3) {
    constexpr int lds = ((10 * 34 * 128 + 255) / 256) * 256 + 8 * 32 * 128;
    static bool attr[XV_MAX_DEVICES] = {false};
    {
      const hipError_t e = xv_allow_dynamic_lds(reinterpret_cast<const void*>(&conv_wgrad_kernel<3>), lds, attr);
      if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(conv_wgrad_kernel<3>, dim3(grid), dim3(256), lds, s, a);
  } else {
    constexpr int lds = 8 * 32 * 128 * 2;
    static bool attr[XV_MAX_DEVICES] = {false};
    {
      const hipError_t e = xv_allow_dynamic_lds(reinterpret_cast<const void*>(&conv_wgrad_kernel<1>), lds, attr);
      if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(conv_wgrad_kernel<1>, dim3(grid), dim3(256), lds, s, a);
  }
  return finish();
}

// conv1_1's filter gradient on the bf16 matrix instruction (conv_first_wgrad_split_kernel): maps that tile exactly in 8x32
// pixels whose X fits 31-bit byte offsets.  grid <= 0: returns the grid it would launch (for the workspace query).
int xv_launch_first_wgrad_split(const float* x, const void* dy, float* dw, float* db, int n, int h, int w, int cin, float* part,
                                int query_only, hipStream_t stream) {
  if (cin < 1 || cin > 4 || (h & 7) || (w & 31) || (int64_t)n * h * w * cin * 4 >= 0x7fff0000LL) return -1;
  const int64_t tiles = (int64_t)n * (h / 8) * (w / 32);
  const int64_t cap = 2 * (int64_t)xv_num_cus();
  const int grid = (int)(tiles < cap ? tiles : cap);
  if (query_only) return grid;
  const int xp = 10 * 34 * cin * 4;
  const int lds = 2 * (8 * 32 * 128 + (xp + 1023) / 1024 * 1024);
#define XV_FWS(C)                                                                                                      \
  {                                                                                                                    \
    static bool attr[XV_MAX_DEVICES] = {false};                                                                        \
    const hipError_t e = xv_allow_dynamic_lds(reinterpret_cast<const void*>(&conv_first_wgrad_split_kernel<C>), lds, attr); \
    if (e != hipSuccess) return -2;                                                                                    \
    hipLaunchKernelGGL(conv_first_wgrad_split_kernel<C>, dim3((unsigned)grid), dim3(512), lds, stream, x, (const __bf16*)dy, dw, \
                       db, n, h, w, part);                                                                             \
  }
  switch (cin) {
    case 1: XV_FWS(1) break;
    case 2: XV_FWS(2) break;
    case 3: XV_FWS(3) break;
    default: XV_FWS(4) break;
  }
#undef XV_FWS
  return grid;
}

// the score layer's filter + bias gradient on the bf16 matrix instruction (score_dense_wgrad_split_kernel): 64 units, C <= 16,
// maps that tile in 8x32 pixels.  Returns the grid (rows of 17 x 64 floats in `part`), < 0 where it does not apply.
int xv_launch_score_wgrad_split(const float* ds, const void* y, int n, int h, int w, int c, float* part, int query_only,
                                hipStream_t stream) {
  if (c < 1 || c > 16 || (h & 7) || (w & 31) || (int64_t)n * h * w * c * 4 >= 0x7fff0000LL) return -1;
  const int64_t tiles = (int64_t)n * (h / 8) * (w / 32);
  const int64_t cap = (int64_t)xv_num_cus();  // (96 KB of LDS: one workgroup per CU)
  const int grid = (int)(tiles < cap ? tiles : cap);
  if (query_only) return grid;
  constexpr int lds = 2 * (8 * 32 * 128 + 8 * 32 * 16 * 4);
  static bool attr[XV_MAX_DEVICES] = {false};
  const hipError_t e = xv_allow_dynamic_lds(reinterpret_cast<const void*>(&score_dense_wgrad_split_kernel), lds, attr);
  if (e != hipSuccess) return -2;
  hipLaunchKernelGGL(score_dense_wgrad_split_kernel, dim3((unsigned)grid), dim3(512), lds, stream, ds, (const __bf16*)y, n, h, w, c, part);
  return grid;
}
