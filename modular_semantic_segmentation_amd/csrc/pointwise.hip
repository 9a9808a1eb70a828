// HBM-bound layers of the FCN for gfx950: first conv (fp32 in), max-pool, bilinear x2 (+add),
// and the fused decoder head (bilinear x8 + relu + 1x1 score + softmax + argmax).
#include <stdlib.h>

#include "xv_common.h"

// Floating-point contraction by the SOURCE only (a * b + c inside one expression), never across statements: the fused
// two-expert head and the unfused path (decoder head -> probability maps -> fusion kernel) must produce the same bits, and
// under the default -ffp-contract=fast the optimizer fuses a product into a later sum wherever the two happen to meet --
// round 4: specialising the fused head on the class count removed a select between `p = e * rsum` and `sum += p`, the
// compiler made it an fma there and not in the kernel that reads p back from memory, and one pixel in a million flipped.
#pragma clang fp contract(on)

namespace {

// ---- conv1_1: relu(conv3x3(x) + b) on the raw fp32 input, fp32 math, bf16 padded-NHWC out -------
// simple_fcn.py:39.  One thread = TWO horizontally adjacent pixels x 64 output channels.  The
// 9*CIN x 64 fp32 weight matrix is read through wave-uniform addresses (scalar loads into SGPRs), each
// weight feeding both pixels; results go through LDS so that every store instruction writes whole
// 128-byte pixel rows.
template <int CIN>
__global__ __launch_bounds__(256) void conv_first_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ b, __bf16* __restrict__ y, int N,
                                                        int H, int W, int relu) {
  const int Wh = W >> 1;  // pixel pairs per row (W is even: multiple of 16)
  const int npair = N * H * Wh;  // < 2^31, checked by the host
  int pr = blockIdx.x * 256 + threadIdx.x;
  pr = pr < npair ? pr : npair - 1;  // tail lanes recompute the last pair; only in-range pixels are stored
  const int row = pr / Wh;         // n * H + py
  const int px = (pr - row * Wh) * 2;
  const int n = row / H;
  const int py = row - n * H;
  float in[3][4][CIN];  // rows py-1..py+1, columns px-1..px+2
#pragma unroll
  for (int dy = 0; dy < 3; ++dy)
#pragma unroll
    for (int dx = 0; dx < 4; ++dx) {
      const int yy = py + dy - 1, xx = px + dx - 1;
      const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
      const float* src = x + (((int64_t)n * H + (ok ? yy : 0)) * W + (ok ? xx : 0)) * CIN;
#pragma unroll
      for (int c = 0; c < CIN; ++c) in[dy][dx][c] = ok ? src[c] : 0.f;
    }
  __shared__ __attribute__((aligned(16))) u32x4 stage[512 * 8];  // [pixel within block][8 slots of 8 channels]
  u32x4* mine = stage + threadIdx.x * 16;
#pragma unroll
  for (int g = 0; g < 4; ++g) {  // 16 output channels at a time (accumulators in VGPRs, weights in SGPRs)
    float a0[16], a1[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) a0[c] = a1[c] = b[g * 16 + c];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) {
          const int t = (dy * 3 + dx) * CIN + ci;
#pragma unroll
          for (int c = 0; c < 16; ++c) {
            const float wv = w[t * 64 + g * 16 + c];
            a0[c] = fmaf(in[dy][dx][ci], wv, a0[c]);
            a1[c] = fmaf(in[dy][dx + 1][ci], wv, a1[c]);
          }
        }
    if (relu) {
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        a0[c] = fmaxf(a0[c], 0.f);
        a1[c] = fmaxf(a1[c], 0.f);
      }
    }
    // 16-byte slot swizzle (slot ^ pixel) keeps the LDS writes of consecutive lanes on distinct banks
    const int p0 = 2 * threadIdx.x, p1 = p0 + 1;
    mine[(2 * g) ^ (p0 & 7)] = u32x4{pack_bf16x2(a0[0], a0[1]), pack_bf16x2(a0[2], a0[3]), pack_bf16x2(a0[4], a0[5]),
                                     pack_bf16x2(a0[6], a0[7])};
    mine[(2 * g + 1) ^ (p0 & 7)] = u32x4{pack_bf16x2(a0[8], a0[9]), pack_bf16x2(a0[10], a0[11]),
                                         pack_bf16x2(a0[12], a0[13]), pack_bf16x2(a0[14], a0[15])};
    mine[8 + ((2 * g) ^ (p1 & 7))] = u32x4{pack_bf16x2(a1[0], a1[1]), pack_bf16x2(a1[2], a1[3]),
                                           pack_bf16x2(a1[4], a1[5]), pack_bf16x2(a1[6], a1[7])};
    mine[8 + ((2 * g + 1) ^ (p1 & 7))] = u32x4{pack_bf16x2(a1[8], a1[9]), pack_bf16x2(a1[10], a1[11]),
                                               pack_bf16x2(a1[12], a1[13]), pack_bf16x2(a1[14], a1[15])};
  }
  // each wave stores its own 128 pixels: no block barrier needed (the wave's LDS region is private)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const int lane = threadIdx.x & 63;
  const int wbase = (threadIdx.x & ~63) * 2;  // first pixel (within the block) of this wave
  const int slot = lane & 7;
  // this lane stores pixel lp = it*8 + (lane>>3) of the wave: pair index advances by 4 per iteration
  int gpair = blockIdx.x * 256 + ((wbase + (lane >> 3)) >> 1);
  int qrow = gpair / Wh;  // n * H + y
  int qxh = gpair - qrow * Wh;
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    const int lp = it * 8 + (lane >> 3);
    if (gpair < npair) {
      const int qn = qrow / H;
      const int qy = qrow - qn * H;
      const u32x4 v = stage[(wbase + lp) * 8 + (slot ^ (lp & 7))];
      *reinterpret_cast<u32x4*>(y + (((int64_t)qn * (H + 2) + (qy + 1)) * (W + 2) + (qxh * 2 + (lp & 1) + 1)) * 64 +
                                slot * 8) = v;
    }
    gpair += 4;
    qxh += 4;
    if (qxh >= Wh) {  // Wh >= 8: at most one row wrap per step
      qxh -= Wh;
      qrow += 1;
    }
  }
}

// ---- conv1_1 on the matrix cores -------------------------------------------------------------------------------------
// The same layer (fp32 image, fp32 weights, fp32 accumulation, one bf16 rounding at the end) as 16 x 16 x 32 bf16 MFMAs:
// every fp32 operand is split EXACTLY into three bf16 terms (x = xh + xm + xl: 3 x 8 significant bits), and the six
// products down to 2^-16 relative (h.h, h.m, m.h, m.m, h.l, l.h) are accumulated in fp32, smallest first -- the dropped
// terms are below 2^-24 of |x||w|, i.e. below the fp32 rounding of the plain FMA chain.  K = 9 CIN <= 27 fits one
// MFMA: k-group g (8 values, lanes 16 g .. 16 g + 15) holds image row dy = g, columns dx = 0..2 x CIN channels in
// memory order (8 of its 9 values for CIN = 3); the three leftover values (dx = 2, ci = 2 of each row) form k-group 3.
// One wave = one tile of 16 consecutive pixels of an image row x 64 channels: 24 MFMAs against 1,728 packed FMAs per
// lane pair of conv_first_kernel, which leaves the layer bound by its output stores.  The tile goes through a 2 KB LDS
// transpose so that each store instruction writes eight whole 128-byte pixel rows (1 KB contiguous).
template <int CIN>
__device__ __forceinline__ bool first_k_map(int g, int e, int& dy, int& dx, int& ci) {
  if (CIN == 3) {
    if (g < 3) {
      dy = g, dx = e / 3, ci = e % 3;
      return true;
    }
    dy = e, dx = 2, ci = 2;
    return e < 3;
  }
  dy = g, dx = e, ci = 0;  // CIN == 1
  return g < 3 && e < 3;
}

// v = h + m + l exactly, each term a bf16 (returned as fp32 bit patterns with zero low halves): truncation keeps the
// top 8 significant bits, the remainder of a 24-bit significand has at most 16, then at most 8
__device__ __forceinline__ void split3_bf16(float v, uint32_t& h, uint32_t& m, uint32_t& l) {
  h = __builtin_bit_cast(uint32_t, v) & 0xffff0000u;
  const float r1 = v - __builtin_bit_cast(float, h);  // exact
  m = __builtin_bit_cast(uint32_t, r1) & 0xffff0000u;
  l = __builtin_bit_cast(uint32_t, r1 - __builtin_bit_cast(float, m));  // exact, fits 8 bits
}
__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) {  // one v_cvt_pk_bf16_f32
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{lo, hi}, bf16x2));
}
__device__ __forceinline__ uint32_t hi16_pair(uint32_t a, uint32_t b) {  // (a >> 16) | (b & 0xffff0000): v_perm_b32
  return __builtin_amdgcn_perm(b, a, 0x07060302u);
}
__device__ __forceinline__ void split3_bf16x8(const float (&v)[8], bf16x8& h, bf16x8& m, bf16x8& l) {
  uint32_t hh[8], mm[8], ll[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) split3_bf16(v[e], hh[e], mm[e], ll[e]);
  h = __builtin_bit_cast(bf16x8, u32x4{hi16_pair(hh[0], hh[1]), hi16_pair(hh[2], hh[3]), hi16_pair(hh[4], hh[5]), hi16_pair(hh[6], hh[7])});
  m = __builtin_bit_cast(bf16x8, u32x4{hi16_pair(mm[0], mm[1]), hi16_pair(mm[2], mm[3]), hi16_pair(mm[4], mm[5]), hi16_pair(mm[6], mm[7])});
  l = __builtin_bit_cast(bf16x8, u32x4{hi16_pair(ll[0], ll[1]), hi16_pair(ll[2], ll[3]), hi16_pair(ll[4], ll[5]), hi16_pair(ll[6], ll[7])});
}

// OUT8: the map is written as e4m3 of value * out_mul (the fp8 graph where conv1_2 takes e4m3 operands, fcn.fp8_plan): a
// tile is then ONE contiguous 1 KB store (16 pixels x 64 channels)
// G7: the map is not written at all; every pixel goes straight to its places in the operand of AdapNet's 7x7 stride-2 conv
// (xv_gather_conv7s2's z [N,H/2,W/2,9*64], adapnet.py:126-127): row variant 0 / 1 of an even row Y at j = Y/2 / Y/2 - 1,
// variant 2 of an odd row at j = (Y-1)/2, columns alike -- 2.25 stores of 128 bytes per pixel on average instead of one,
// and neither the 0.6 GB map nor the gather's read of it.  Positions of z without a source pixel (variant 1 in the last
// row / column) are never written: the caller's buffer holds zeros there (as xv_gather_conv7s2 leaves them).
template <int CIN, bool OUT8 = false, bool G7 = false>
__global__ __launch_bounds__(256) void conv_first_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ b, __bf16* __restrict__ y, int N,
                                                             int H, int W, int relu, int tpw, float out_mul = 1.f) {
  static_assert(!(OUT8 && G7), "the gathered form writes bf16");
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15, g = lane >> 4;
  const int Wt = W >> 4;  // tiles per image row (W is a multiple of 16, checked by the host)
  const int ntiles = N * H * Wt;
  __shared__ __attribute__((aligned(16))) u32x4 stage_all[4 * 128];
  u32x4* stage = stage_all + wave * 128;  // [16 pixels][8 slots of 8 channels], private to the wave

  // weight fragments (A operand: row = channel j of the 16-channel block, k-group g), split three ways.  The bias
  // rides in the first spare k slot of group 3 against a constant 1.0 on the image side (1.0 = xh exactly, so the
  // three terms wh + wm + wl = b enter the sum exactly): the accumulators start from zero and cost no registers.
  constexpr int BIAS_E = CIN == 3 ? 3 : 0;
  bf16x8 wh[4], wm[4], wl[4];
#pragma unroll
  for (int jb = 0; jb < 4; ++jb) {
    float wv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      int dy, dx, ci;
      const bool ok = first_k_map<CIN>(g, e, dy, dx, ci);
      wv[e] = ok ? w[((dy * 3 + dx) * CIN + ci) * 64 + jb * 16 + j] : 0.f;
      if (e == BIAS_E) wv[e] = g == 3 ? b[jb * 16 + j] : wv[e];
    }
    split3_bf16x8(wv, wh[jb], wm[jb], wl[jb]);
  }

  // Lane constants of the tap loads.  A lane of k-group g < 3 reads image row py + g - 1 at columns px - 1, px, px + 1
  // (CIN contiguous floats each); a lane of k-group 3 reads column px + 1 of rows py - 1, py, py + 1 (CIN == 3 only: it
  // uses the last channel).  Position t is at float offset (tile origin) + lc + t * lstride; it lies outside the image
  // when one of the tile's edge flags (bit 0 top row, 1 bottom row, 2 first tile of the row, 3 last tile; bit 4: set
  // always, kills the lanes that never load; bit 5: there is no such tile, kills every lane) meets the position's
  // kill mask.
  const bool main = g < 3;
  const int lc = main ? ((g - 1) * W + j - 1) * CIN : (-W + j + 1) * CIN;
  const int lstride = main ? CIN : W * CIN;
  int kill[3];
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const int dy = main ? g : t, dx = main ? t : 2;
    kill[t] = 32 | (dy == 0 ? 1 : 0) | (dy == 2 ? 2 : 0) | ((j == 0 && dx == 0) ? 4 : 0) | ((j == 15 && dx == 2) ? 8 : 0) |
              ((CIN == 1 && !main) ? 16 : 0);
  }
  const int slot = ((g & 1) << 1) | (g >> 1);
  const int st_w0 = j * 8 + (slot ^ (j & 7)), st_w1 = j * 8 + ((slot + 4) ^ (j & 7));
  const int pj0 = lane >> 3, sl = lane & 7;
  const int st_r0 = pj0 * 8 + (sl ^ (pj0 & 7)), st_r1 = (pj0 + 8) * 8 + (sl ^ (pj0 & 7));
  const int st_g = pj0 * 64 + sl * 8;

  // wave-uniform tile walk (no divisions in the loop): tile -> (image n, row py, tile tx of the row)
  // a wave takes a contiguous run of tpw tiles (its stores and taps stream through memory; measured 5-10 % ahead of a
  // grid-stride assignment); tpw = 0: grid stride
  const int wstride = tpw ? 1 : gridDim.x * 4;
  int tile = tpw ? (blockIdx.x * 4 + wave) * tpw : blockIdx.x * 4 + wave;
  const int tile_end = tpw ? (tile + tpw < ntiles ? tile + tpw : ntiles) : ntiles;
  int n, py, tx;
  {
    const int row = tile / Wt;
    tx = tile - row * Wt;
    n = row / H;
    py = row - n * H;
  }
  const int drow = wstride / Wt, dtx = wstride - drow * Wt;
  const int dn = drow / H, dpy = drow - dn * H;

  // requests the taps of tile TILE = (n, py, tx), unmasked: they are masked where they are consumed, one tile later, so
  // that nothing here waits for the loads
#define XV_FIRST_LOAD(TILE)                                                                                           \
  {                                                                                                                   \
    const int sbase = ((n * H + py) * W + tx * 16) * CIN;                                                             \
    const int edge = (py == 0 ? 1 : 0) | (py == H - 1 ? 2 : 0) | (tx == 0 ? 4 : 0) | (tx == Wt - 1 ? 8 : 0) | 16 |    \
                     ((TILE) < tile_end ? 0 : 32);                                                                      \
    _Pragma("unroll") for (int t = 0; t < 3; ++t) {                                                                   \
      ok[t] = (kill[t] & edge) == 0;                                                                                  \
      const int off = ok[t] ? sbase + lc + t * lstride : 0;                                                           \
      _Pragma("unroll") for (int c = 0; c < CIN; ++c) raw[t][c] = x[off + c];                                         \
    }                                                                                                                 \
  }

  const uint32_t floor2 = relu ? 0u : 0x80008000u;  // relu as a packed signed-integer max (xv_common.h); -32768 = none
  float raw[3][CIN];
  bool ok[3];
  XV_FIRST_LOAD(tile)
  for (; tile < tile_end; tile += wstride) {
    // B operand of this tile (column = pixel j, k-group g): masked taps in k order, split three ways
    float v[8];
    if (CIN == 3) {
      v[0] = ok[0] ? (main ? raw[0][0] : raw[0][2]) : 0.f;
      v[1] = main ? (ok[0] ? raw[0][1] : 0.f) : (ok[1] ? raw[1][2] : 0.f);
      v[2] = main ? (ok[0] ? raw[0][2] : 0.f) : (ok[2] ? raw[2][2] : 0.f);
      v[3] = main ? (ok[1] ? raw[1][0] : 0.f) : 1.f;  // k-group 3: the bias slot
      v[4] = main && ok[1] ? raw[1][1] : 0.f;
      v[5] = main && ok[1] ? raw[1][2] : 0.f;
      v[6] = main && ok[2] ? raw[2][0] : 0.f;
      v[7] = main && ok[2] ? raw[2][1] : 0.f;
    } else {
      v[0] = main ? (ok[0] ? raw[0][0] : 0.f) : 1.f;  // k-group 3: the bias slot
      v[1] = ok[1] ? raw[1][0] : 0.f, v[2] = ok[2] ? raw[2][0] : 0.f;
      v[3] = v[4] = v[5] = v[6] = v[7] = 0.f;
    }
    bf16x8 xh, xm, xl;
    split3_bf16x8(v, xh, xm, xl);
    __bf16* dst = y + (((int64_t)n * (H + 2) + (py + 1)) * (W + 2) + (tx * 16 + 1)) * 64 + st_g;
    char* dst8 = reinterpret_cast<char*>(y) + (((int64_t)n * (H + 2) + (py + 1)) * (W + 2) + (tx * 16 + 1)) * 64 + lane * 16;
    // G7: the lane's pixel X = 16 tx + pj0 (and X + 8: four operand columns further) in row variant A = 0 (even py) / 2 (odd)
    // and column variant C = 0 (even X) / 2 (odd), at operand pixel (py >> 1, X >> 1)
    char* zA_C = nullptr;
    bool g7_d0 = false, g7_rowb = false;
    if constexpr (G7) {
      const int Ho = H >> 1, Wo = W >> 1, X = tx * 16 + pj0;
      const int rvA = (py & 1) ? 2 : 0, cvC = (pj0 & 1) ? 2 : 0;
      zA_C = reinterpret_cast<char*>(y) + ((((int64_t)n * (Ho + 2) + (py >> 1) + 1) * (Wo + 2) + (X >> 1) + 1) * 576 + (rvA * 3 + cvC) * 64) * 2 + sl * 16;
      g7_d0 = X >= 2;                        // the even pixel's second column place (variant 1 at X/2 - 1) exists
      g7_rowb = (py & 1) == 0 && py >= 2;    // the even row's second place (variant 1 at py/2 - 1) exists
    }
    // next tile: advance the walk and request its taps; they land behind this tile's MFMAs, and this tile's stores are
    // issued after them (the vector-memory counter retires in order: waiting for the taps then never waits for the
    // stores issued behind them)
    {
      tx += dtx;
      const int c1 = tx >= Wt ? 1 : 0;
      tx -= c1 ? Wt : 0;
      py += dpy + c1;
      const int c2 = py >= H ? 1 : 0;
      py -= c2 ? H : 0;
      n += dn + c2;
    }
    XV_FIRST_LOAD(tile + wstride)
    __builtin_amdgcn_sched_barrier(0x78f);  // no memory request may sink below the MFMAs (everything else may move)
    f32x4 acc[4];
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) acc[jb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[jb], xl, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) acc[jb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[jb], xh, acc[jb], 0, 0, 0);
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) acc[jb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm[jb], xm, acc[jb], 0, 0, 0);
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) acc[jb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[jb], xm, acc[jb], 0, 0, 0);
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) acc[jb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm[jb], xh, acc[jb], 0, 0, 0);
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) acc[jb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[jb], xh, acc[jb], 0, 0, 0);
    if constexpr (OUT8) {
      // lane (pixel j, row group g) holds channels 16 jb + 4 g .. + 3 of block jb: one dword of e4m3 each, at 16-byte slot
      // jb ^ ((j >> 1) & 3) of the pixel's 64-byte row in the wave's stage (two-way write conflicts: free); read back as
      // 16 bytes per lane = pixel lane >> 2, slot lane & 3
      uint32_t* st32 = reinterpret_cast<uint32_t*>(stage);
#pragma unroll
      for (int jb = 0; jb < 4; ++jb) {
        f32x4 t = acc[jb];
        if (relu) t = f32x4{fmaxf(t.x, 0.f), fmaxf(t.y, 0.f), fmaxf(t.z, 0.f), fmaxf(t.w, 0.f)};
        st32[j * 16 + ((jb ^ ((j >> 1) & 3)) << 2) + g] = xv_pack_fp8x4(t.x, t.y, t.z, t.w, out_mul);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const int pj = lane >> 2, sl = lane & 3;
      const u32x4 r = stage[pj * 4 + (sl ^ ((pj >> 1) & 3))];
      *reinterpret_cast<u32x4*>(dst8) = r;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      continue;
    }
    u32x2 packed[4];
#pragma unroll
    for (int jb = 0; jb < 4; ++jb)
      packed[jb] = u32x2{pk_max_i16(cvt_pk_bf16(acc[jb][0], acc[jb][1]), floor2),
                         pk_max_i16(cvt_pk_bf16(acc[jb][2], acc[jb][3]), floor2)};
    // lane (pixel j, row group g) now holds 8 consecutive channels of each 32-channel pair, starting at channel
    // {0, 16, 8, 24}[g] of the pair: 16-byte slot {0, 2, 1, 3}[g] + 4 pair of the pixel's 128-byte row
    u32x4 o0, o1;
    xv_pair16(packed[0], packed[1], o0);
    xv_pair16(packed[2], packed[3], o1);
    stage[st_w0] = o0;
    stage[st_w1] = o1;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const u32x4 r0 = stage[st_r0], r1 = stage[st_r1];
    if constexpr (G7) {
      const int Wo = W >> 1;
      const int64_t rowb = -(int64_t)(Wo + 2) * 1152 + 3 * 128;   // row variant 0 -> 1, one operand row up
      const bool even_x = (pj0 & 1) == 0;
      *reinterpret_cast<u32x4*>(zA_C) = r0;
      *reinterpret_cast<u32x4*>(zA_C + 4 * 1152) = r1;
      if (even_x) {                                                // column variant 0 -> 1, one operand column to the left
        if (g7_d0) *reinterpret_cast<u32x4*>(zA_C - 1152 + 128) = r0;
        *reinterpret_cast<u32x4*>(zA_C + 3 * 1152 + 128) = r1;
      }
      if (g7_rowb) {
        *reinterpret_cast<u32x4*>(zA_C + rowb) = r0;
        *reinterpret_cast<u32x4*>(zA_C + rowb + 4 * 1152) = r1;
        if (even_x) {
          if (g7_d0) *reinterpret_cast<u32x4*>(zA_C + rowb - 1152 + 128) = r0;
          *reinterpret_cast<u32x4*>(zA_C + rowb + 3 * 1152 + 128) = r1;
        }
      }
    } else {
      *reinterpret_cast<u32x4*>(dst) = r0;
      *reinterpret_cast<u32x4*>(dst + 8 * 64) = r1;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
#undef XV_FIRST_LOAD
}

__device__ inline u32x4 bf16x8_max(u32x4 a, u32x4 b) {
  u32x4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float alo = bf16_bits_to_f32(a[i] & 0xffffu), ahi = __builtin_bit_cast(float, a[i] & 0xffff0000u);
    const float blo = bf16_bits_to_f32(b[i] & 0xffffu), bhi = __builtin_bit_cast(float, b[i] & 0xffff0000u);
    const uint32_t lo = __builtin_bit_cast(uint32_t, fmaxf(alo, blo)) >> 16;
    const uint32_t hi = __builtin_bit_cast(uint32_t, fmaxf(ahi, bhi)) & 0xffff0000u;
    r[i] = lo | hi;
  }
  return r;
}

// ---- max_pooling2d(2,2) on padded-NHWC bf16; one thread = 8 channels of one output pixel --------
__global__ __launch_bounds__(256) void maxpool_kernel(const __bf16* __restrict__ x, __bf16* __restrict__ y, int N,
                                                     int Ho, int Wo, int C) {
  const int c8 = C >> 3;
  const int64_t total = (int64_t)N * Ho * Wo * c8;
  const int Hi = Ho * 2, Wi = Wo * 2;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int cg = (int)(idx % c8);
    int64_t r = idx / c8;
    const int ox = (int)(r % Wo);
    r /= Wo;
    const int oy = (int)(r % Ho);
    const int n = (int)(r / Ho);
    const __bf16* src = x + (((int64_t)n * (Hi + 2) + (2 * oy + 1)) * (Wi + 2) + (2 * ox + 1)) * C + cg * 8;
    const int64_t rowp = (int64_t)(Wi + 2) * C;
    u32x4 v = bf16x8_max(*reinterpret_cast<const u32x4*>(src), *reinterpret_cast<const u32x4*>(src + C));
    v = bf16x8_max(v, *reinterpret_cast<const u32x4*>(src + rowp));
    v = bf16x8_max(v, *reinterpret_cast<const u32x4*>(src + rowp + C));
    *reinterpret_cast<u32x4*>(y + (((int64_t)n * (Ho + 2) + (oy + 1)) * (Wo + 2) + (ox + 1)) * C + cg * 8) = v;
  }
}

// Bilinear transposed-conv taps ([TF1] conv2d_transpose 'same': out o receives in i through kernel
// index p with o = i*S + p - S/2, k = 2S): sources i1 = (o + S/2) / S with p1 = (o + S/2) % S and
// i0 = i1 - 1 with p0 = p1 + S; 1-D weight w1[p] = 1 - |p/S - (2S-1-S%2)/(2S)| (custom_layers.py:15-21).
// Out-of-range sources fall on the zero border of the padded layout and contribute exactly 0.
template <int S>
__device__ inline void bilinear_taps(int o, int& i1, float& w_i1, float& w_i0) {
  const int t = o + S / 2;
  i1 = t / S;
  const int p1 = t - i1 * S;
  constexpr float center = (2.f * S - 1.f - (S % 2)) / (2.f * S);
  w_i1 = 1.f - fabsf((float)p1 / S - center);
  w_i0 = 1.f - fabsf((float)(p1 + S) / S - center);
}

// ---- upscore_conv5 + add_score: y = residual + relu(bilinear_x2(x)) (simple_fcn.py:82-85) ------
// scale / shift (may be null): inference batch norm between the deconv and its relu (custom_layers.py:112-119)
__global__ __launch_bounds__(256) void upsample2x_kernel(const __bf16* __restrict__ x, const __bf16* __restrict__ res,
                                                        const float* __restrict__ scale, const float* __restrict__ shift,
                                                        __bf16* __restrict__ y, int N, int Hi, int Wi, int C,
                                                        int relu) {
  const int c8 = C >> 3;
  const int Ho = Hi * 2, Wo = Wi * 2;
  const int64_t total = (int64_t)N * Ho * Wo * c8;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int cg = (int)(idx % c8);
    int64_t r = idx / c8;
    const int ox = (int)(r % Wo);
    r /= Wo;
    const int oy = (int)(r % Ho);
    const int n = (int)(r / Ho);
    int iy1, ix1;
    float wy1, wy0, wx1, wx0;
    bilinear_taps<2>(oy, iy1, wy1, wy0);
    bilinear_taps<2>(ox, ix1, wx1, wx0);
    // padded coords: logical i -> i + 1; i0 = i1 - 1 >= -1 and i1 <= Hi are inside the padded buffer
    const __bf16* p00 = x + (((int64_t)n * (Hi + 2) + iy1) * (Wi + 2) + ix1) * C + cg * 8;  // (iy0, ix0)
    const int64_t rowp = (int64_t)(Wi + 2) * C;
    const u32x4 a00 = *reinterpret_cast<const u32x4*>(p00), a01 = *reinterpret_cast<const u32x4*>(p00 + C);
    const u32x4 a10 = *reinterpret_cast<const u32x4*>(p00 + rowp), a11 = *reinterpret_cast<const u32x4*>(p00 + rowp + C);
    const float w00 = wy0 * wx0, w01 = wy0 * wx1, w10 = wy1 * wx0, w11 = wy1 * wx1;
    u32x4 rv = u32x4{0u, 0u, 0u, 0u};
    const int64_t oidx = (((int64_t)n * (Ho + 2) + (oy + 1)) * (Wo + 2) + (ox + 1)) * C + cg * 8;
    if (res) rv = *reinterpret_cast<const u32x4*>(res + oidx);
    u32x4 out;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float v[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int sh = h * 16;
        const float f00 = bf16_bits_to_f32((a00[i] >> sh) & 0xffffu), f01 = bf16_bits_to_f32((a01[i] >> sh) & 0xffffu);
        const float f10 = bf16_bits_to_f32((a10[i] >> sh) & 0xffffu), f11 = bf16_bits_to_f32((a11[i] >> sh) & 0xffffu);
        float u = f00 * w00 + f01 * w01 + f10 * w10 + f11 * w11;
        if (scale) u = u * scale[cg * 8 + i * 2 + h] + shift[cg * 8 + i * 2 + h];
        if (relu) u = fmaxf(u, 0.f);
        v[h] = u + bf16_bits_to_f32((rv[i] >> sh) & 0xffffu);
      }
      out[i] = pack_bf16x2(v[0], v[1]);
    }
    *reinterpret_cast<u32x4*>(y + oidx) = out;
  }
}

// ---- decoder head: bilinear x8 + relu + 1x1 score + softmax + argmax ----------------------------
// simple_fcn.py:129-133 + basic_fusion_model.py:21-22.
//
// `fused` = relu(score_conv4) + relu(bilinear_x2(..)) is non-negative by construction and the
// bilinear weights are positive, so relu(bilinear_x8(fused)) == bilinear_x8(fused): the x8 deconv and
// the 1x1 `score` conv are both linear and commute.  The head therefore runs the 1x1 conv at 1/8
// resolution (U -> C channels on h*w pixels instead of 64*h*w) and interpolates C class scores instead
// of U features: 16x fewer FMAs per output pixel, same value up to fp32 summation order.  The bias is
// added after the interpolation (at the image border the zero-padded bilinear weights do not sum to 1).
//
// Kernel 1: S[n][i][j][k] = sum_u fused[n,i,j,u] * Ws[u][k] into a zero-bordered fp32 [N][h+2][w+2][CP]
// workspace (CP = C rounded up to 4).  Score weights sit zero-padded in LDS and are read with wave-uniform
// (broadcast) addresses.
template <int CM>
__global__ __launch_bounds__(128) void score_lowres_kernel(const __bf16* __restrict__ f, const float* __restrict__ ws_g,
                                                          int N, int Hi, int Wi, int U, int C, float* __restrict__ S) {
  extern __shared__ __attribute__((aligned(16))) float wsm[];  // [U][CM], zero padded
  for (int i = threadIdx.x; i < U * CM; i += 128) {
    const int u = i / CM, k = i - u * CM;
    wsm[i] = k < C ? ws_g[u * C + k] : 0.f;
  }
  __syncthreads();
  const int64_t total = (int64_t)N * (Hi + 2) * (Wi + 2);
  const int64_t pp = (int64_t)blockIdx.x * 128 + threadIdx.x;  // padded pixel index (same geometry as `fused`)
  if (pp >= total) return;
  const int x = (int)(pp % (Wi + 2));
  const int y = (int)((pp / (Wi + 2)) % (Hi + 2));
  float sc[CM];
#pragma unroll
  for (int k = 0; k < CM; ++k) sc[k] = 0.f;
  const bool interior = x >= 1 && x <= Wi && y >= 1 && y <= Hi;
  if (interior) {
    const __bf16* src = f + pp * U;
    for (int u0 = 0; u0 < U; u0 += 8) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(src + u0);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float fv = bf16_bits_to_f32((v[i >> 1] >> ((i & 1) * 16)) & 0xffffu);
        const float* wrow = wsm + (u0 + i) * CM;  // wave-uniform address: LDS broadcast reads
#pragma unroll
        for (int k4 = 0; k4 < CM; k4 += 4) {
          const f32x4 wv = *reinterpret_cast<const f32x4*>(wrow + k4);
          sc[k4] = fmaf(fv, wv.x, sc[k4]);
          sc[k4 + 1] = fmaf(fv, wv.y, sc[k4 + 1]);
          sc[k4 + 2] = fmaf(fv, wv.z, sc[k4 + 2]);
          sc[k4 + 3] = fmaf(fv, wv.w, sc[k4 + 3]);
        }
      }
    }
  }
  float* dst = S + pp * CM;
#pragma unroll
  for (int k4 = 0; k4 < CM; k4 += 4) *reinterpret_cast<f32x4*>(dst + k4) = f32x4{sc[k4], sc[k4 + 1], sc[k4 + 2], sc[k4 + 3]};
}

// Kernel 2: one thread per output pixel: 4-tap bilinear interpolation of the CM low-resolution class
// scores, + bias, softmax, argmax.  The per-pixel pieces are device functions shared with the fused two-expert head
// (fused_head_kernel), so both paths execute the same arithmetic in the same order: their labels are bit-identical.
// the four low-resolution score vectors around output pixel (oy, ox) (shared by the 4 output pixels ox = 4m .. 4m + 3:
// bilinear_taps<8> changes its source column at ox = 4 mod 8 only)
template <int CM>
__device__ __forceinline__ void head_load_taps(const float* __restrict__ S, int n, int iy1, int ix1, int Hi, int Wi,
                                               f32x4 (&a)[CM / 4], f32x4 (&b)[CM / 4], f32x4 (&c)[CM / 4], f32x4 (&d)[CM / 4]) {
  // padded coords: logical source (iy1-1, ix1-1) is padded (iy1, ix1)
  const float* p00 = S + (((int64_t)n * (Hi + 2) + iy1) * (Wi + 2) + ix1) * CM;
  const int64_t rowp = (int64_t)(Wi + 2) * CM;
#pragma unroll
  for (int k4 = 0; k4 < CM / 4; ++k4) {
    a[k4] = *reinterpret_cast<const f32x4*>(p00 + 4 * k4);
    b[k4] = *reinterpret_cast<const f32x4*>(p00 + CM + 4 * k4);
    c[k4] = *reinterpret_cast<const f32x4*>(p00 + rowp + 4 * k4);
    d[k4] = *reinterpret_cast<const f32x4*>(p00 + rowp + CM + 4 * k4);
  }
}

// logits of one output pixel from its taps: explicit fmaf chain (shared by the unfused and the fused head: identical
// bits by construction), then the bias
template <int CM>
__device__ __forceinline__ void head_eval_taps(const f32x4 (&a)[CM / 4], const f32x4 (&b)[CM / 4], const f32x4 (&c)[CM / 4],
                                               const f32x4 (&d)[CM / 4], float wy1, float wy0, float wx1, float wx0,
                                               const float* __restrict__ bs_g, int C, float (&sc)[CM]) {
  const float w00 = wy0 * wx0, w01 = wy0 * wx1, w10 = wy1 * wx0, w11 = wy1 * wx1;
#pragma unroll
  for (int k4 = 0; k4 < CM / 4; ++k4) {
    sc[4 * k4] = fmaf(d[k4].x, w11, fmaf(c[k4].x, w10, fmaf(b[k4].x, w01, a[k4].x * w00)));
    sc[4 * k4 + 1] = fmaf(d[k4].y, w11, fmaf(c[k4].y, w10, fmaf(b[k4].y, w01, a[k4].y * w00)));
    sc[4 * k4 + 2] = fmaf(d[k4].z, w11, fmaf(c[k4].z, w10, fmaf(b[k4].z, w01, a[k4].z * w00)));
    sc[4 * k4 + 3] = fmaf(d[k4].w, w11, fmaf(c[k4].w, w10, fmaf(b[k4].w, w01, a[k4].w * w00)));
  }
#pragma unroll
  for (int k = 0; k < CM; ++k) sc[k] += bs_g[k < C ? k : C - 1];
}

template <int CM>
__device__ __forceinline__ void head_logits(const float* __restrict__ S, const float* __restrict__ bs_g, int n, int oy, int ox,
                                            int Hi, int Wi, int C, float (&sc)[CM]) {
  int iy1, ix1;
  float wy1, wy0, wx1, wx0;
  bilinear_taps<8>(oy, iy1, wy1, wy0);
  bilinear_taps<8>(ox, ix1, wx1, wx0);
  f32x4 a[CM / 4], b[CM / 4], c[CM / 4], d[CM / 4];
  head_load_taps<CM>(S, n, iy1, ix1, Hi, Wi, a, b, c, d);
  head_eval_taps<CM>(a, b, c, d, wy1, wy0, wx1, wx0, bs_g, C, sc);
}

template <int CM>
__device__ __forceinline__ float head_max(const float (&sc)[CM], int C) {
  float m = sc[0];
#pragma unroll
  for (int k = 1; k < CM; ++k)
    if (k < C) m = fmaxf(m, sc[k]);
  return m;
}

// labels only (the experts of a Bayes fusion): argmax(softmax(x)) is argmax(x) unless the runner-up is so close that
// the two probabilities round to the same float (|difference| < ~1e-7); only then does the reference's tie rule
// (lowest index among equal PROBABILITIES) need the probabilities themselves.  Returns -1 in that case.
template <int CM>
__device__ __forceinline__ int head_label_fast(const float (&sc)[CM], float m, int C) {
  // exactly one k with (m - sc[k]) <= 1e-5 (the maximum itself)  <=>  the SECOND largest value, duplicates counted, is more
  // than 1e-5 below m (the subtraction is monotone in sc[k]): s2 = med3(s1, s2, x) under the running maximum s1.  Counting the
  // near classes cost a subtraction, a compare and an add per class: 2 242 -> 1 822 vector instructions in the fused Bayes head.
  float s1 = sc[0], s2 = -__builtin_inff();
  int bi = 0;
#pragma unroll
  for (int k = 1; k < CM; ++k)
    if (k < C) {
      s2 = __builtin_amdgcn_fmed3f(s1, s2, sc[k]);
      s1 = fmaxf(s1, sc[k]);
    }
#pragma unroll
  for (int k = CM - 1; k >= 0; --k)
    if (k < C && sc[k] == m) bi = k;
  return (m - s2) <= 1e-5f ? -1 : bi;
}

// sc <- softmax(sc) (tf.nn.softmax: exp(x - max) / sum); returns the label, lowest index on ties
template <int CM>
__device__ __forceinline__ int head_softmax(float (&sc)[CM], float m, int C) {
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < CM; ++k) {
    sc[k] = k < C ? xv_fast_exp(sc[k] - m) : 0.f;
    sum += sc[k];
  }
  const float rsum = xv_fast_rcp(sum);
  float best = -1.f;
  int bi = 0;
#pragma unroll
  for (int k = 0; k < CM; ++k) {
    sc[k] = sc[k] * rsum;
    if (k < C && sc[k] > best) {
      best = sc[k];
      bi = k;
    }
  }
  return bi;
}

template <int CM>
__global__ __launch_bounds__(256) void decoder_head_kernel(const float* __restrict__ S, const float* __restrict__ bs_g,
                                                          int N, int Hi, int Wi, int C, float* __restrict__ score,
                                                          float* __restrict__ prob, int64_t* __restrict__ label) {
  const int Ho = Hi * 8, Wo = Wi * 8;
  const int64_t npix = (int64_t)N * Ho * Wo;
  const int64_t opix = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (opix >= npix) return;
  const int ox = (int)(opix % Wo);
  const int oy = (int)((opix / Wo) % Ho);
  const int n = (int)(opix / ((int64_t)Wo * Ho));
  float sc[CM];
  head_logits<CM>(S, bs_g, n, oy, ox, Hi, Wi, C, sc);
  if (score) {
#pragma unroll
    for (int k = 0; k < CM; ++k)
      if (k < C) score[opix * C + k] = sc[k];
  }
  if (prob || label) {
    const float m = head_max<CM>(sc, C);
    if (!prob) {
      const int fast = head_label_fast<CM>(sc, m, C);
      if (fast >= 0) {
        label[opix] = fast;
        return;
      }
    }
    const int bi = head_softmax<CM>(sc, m, C);
    if (prob) {
#pragma unroll
      for (int k = 0; k < CM; ++k)
        if (k < C) prob[opix * C + k] = sc[k];
    }
    if (label) label[opix] = bi;
  }
}

// decoder_head_kernel for the label alone, FOUR consecutive output pixels ox = 4m .. 4m + 3 per thread: they share their four
// low-resolution source vectors (bilinear_taps<8> changes its source column at ox = 4 mod 8 only) and leave as two 16-byte
// stores -- the same device functions per pixel, so the same labels.  16 images of 768x384, score_lowres + head: 44 -> 33 us
// (one pixel per thread issued 12 16-byte loads for every 8-byte result).
template <int CM>
__global__ __launch_bounds__(256) void decoder_head_label4_kernel(const float* __restrict__ S, const float* __restrict__ bs_g,
                                                                 int N, int Hi, int Wi, int C, int64_t* __restrict__ label) {
  const int Ho = Hi * 8, Wo = Wi * 8, Wq = Wo / 4;
  const int64_t nquads = (int64_t)N * Ho * Wq;
  const int64_t quad = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (quad >= nquads) return;
  const int ox0 = (int)(quad % Wq) * 4;
  const int oy = (int)((quad / Wq) % Ho);
  const int n = (int)(quad / ((int64_t)Wq * Ho));
  int iy1, ix1;
  float wy1, wy0;
  bilinear_taps<8>(oy, iy1, wy1, wy0);
  {
    float u1, u0;
    bilinear_taps<8>(ox0, ix1, u1, u0);
  }
  f32x4 ta[CM / 4], tb[CM / 4], tc[CM / 4], td[CM / 4];
  head_load_taps<CM>(S, n, iy1, ix1, Hi, Wi, ta, tb, tc, td);
  int64_t out[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    int ixp;
    float wx1, wx0;
    bilinear_taps<8>(ox0 + p, ixp, wx1, wx0);
    float sc[CM];
    head_eval_taps<CM>(ta, tb, tc, td, wy1, wy0, wx1, wx0, bs_g, C, sc);
    const float m = head_max<CM>(sc, C);
    int l = head_label_fast<CM>(sc, m, C);
    if (l < 0) l = head_softmax<CM>(sc, m, C);
    out[p] = l;
  }
  typedef __attribute__((ext_vector_type(2))) long long i64x2;
  int64_t* dst = label + quad * 4;
  *reinterpret_cast<i64x2*>(dst) = i64x2{out[0], out[1]};
  *reinterpret_cast<i64x2*>(dst + 2) = i64x2{out[2], out[3]};
}

// ---- fused two-expert head: both experts' low-resolution class scores -> per-pixel logits -> softmax / argmax of
// each expert -> Bayes (bayes_mix.py:33-58) or Dirichlet (dirichlet_mix.py:14-36,96-136) fusion -> ONE fused label.
// Replaces, for the default prediction of a two-expert fusion model, two decoder_head launches + the fusion kernel and
// every per-pixel intermediate between them (Bayes: two int64 label maps written and read back; Dirichlet: two fp32
// probability maps, 2 x 4C B per pixel each way).  The fusion arithmetic is the one of fusion.hip's kernels, term for
// term, on the values the unfused path would have stored.  Tables in LDS: tab [2][C][CM], lognorm [2][CM] (Dirichlet),
// logprior [CM], dec [C][C] (Bayes: the decision per label pair, built by the workgroup).
// FULL: the class count IS CM (the 12 classes of the headline model): every `k < C` folds away -- 165 of the Dirichlet
// form's ~800 vector instructions per pixel were compare-selects on the run-time class count.
template <int CM, int DIRICHLET, bool FULL = false, int P = (DIRICHLET ? 1 : 4)>
__global__ __launch_bounds__(256) void fused_head_kernel(const float* __restrict__ Sa, const float* __restrict__ Sb,
                                                        const float* __restrict__ ba, const float* __restrict__ bb, int N,
                                                        int Hi, int Wi, int C_, const float* __restrict__ tab_g,
                                                        const float* __restrict__ lognorm_g,
                                                        const float* __restrict__ logprior_g, int64_t* __restrict__ fused) {
  const int C = FULL ? CM : C_;
  extern __shared__ __attribute__((aligned(16))) float tab[];
  float* ln = tab + (DIRICHLET ? 2 * CM * CM : 2 * C * CM);
  float* lp = ln + 2 * CM;
  if constexpr (DIRICHLET) {
    // transposed: tab[(e CM + k) CM + c] = alpha_e[c][k] - 1, so that the twelve dot products of a pixel advance together,
    // two classes per v_pk_fma_f32 (below)
    for (int i = threadIdx.x; i < 2 * CM * CM; i += 256) {
      const int c = i % CM, k = (i / CM) % CM, e = i / (CM * CM);
      tab[i] = (c < C && k < C) ? tab_g[(e * C + c) * C + k] : 0.f;
    }
  } else {
    for (int i = threadIdx.x; i < 2 * C * CM; i += 256) {
      const int k = i % CM, row = i / CM;
      tab[i] = k < C ? tab_g[row * C + k] : 0.f;
    }
  }
  for (int i = threadIdx.x; i < 2 * CM; i += 256) {
    const int k = i % CM, e = i / CM;
    ln[i] = (DIRICHLET && k < C) ? lognorm_g[e * C + k] : 0.f;
  }
  if (threadIdx.x < CM) lp[threadIdx.x] = threadIdx.x < C ? logprior_g[threadIdx.x] : 0.f;
  __syncthreads();
  // Bayes: the fused label is a function of the two experts' labels alone -- dec[a][b] = argmax_k (tab_0[a][k] + tab_1[b][k] +
  // logprior[k]), built here by the workgroup with the sums in the order of bayes_fuse_kernel (fusion.hip; as bayes_fuse2_kernel
  // does for the unfused path): a pixel is then ONE 4-byte LDS lookup instead of six lane-varying 16-byte row reads, 36 adds
  // and a 12-way argmax chain, and the thread keeps two labels per pixel instead of CM sums.
  int* dec = reinterpret_cast<int*>(lp + CM);  // [C][C]
  if constexpr (!DIRICHLET) {
    for (int i = threadIdx.x; i < C * C; i += 256) {
      const float* ra = tab + (i / C) * CM;
      const float* rb = tab + (C + i % C) * CM;
      float best = 0.f;
      int bi = 0;
      for (int k = 0; k < C; ++k) {
        float sc = ra[k];
        sc = sc + rb[k];
        const float v = sc + lp[k];
        if (k == 0 || v > best) {
          best = v;
          bi = k;
        }
      }
      dec[i] = bi;
    }
    __syncthreads();
  }
  // Bayes: one thread = P = FOUR consecutive output pixels ox = 4m .. 4m + 3: they share their four low-resolution source
  // vectors (24 16-byte loads per expert pair instead of 96) and leave as two 16-byte stores: 71.8 -> 54.6 us for 37.7 MB
  // at 16 images of 768x384 (profiles/r3_elementwise.json: not HBM-bound, 0.10 of the HBM rate: the per-pixel argmax
  // chains and the lane-varying table reads in LDS remain).  Dirichlet (24 logs + 288 FMAs per pixel, 48 more live
  // registers per extra pixel) keeps one pixel per thread in this scalar form: four measured 14 % slower.  (C == CM runs
  // fused_dirichlet_head_pk_kernel below.)
  const int Ho = Hi * 8, Wo = Wi * 8, Wq = Wo / P;
  const int64_t nquads = (int64_t)N * Ho * Wq;
  const int64_t quad = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (quad >= nquads) return;
  const int ox0 = (int)(quad % Wq) * P;
  const int oy = (int)((quad / Wq) % Ho);
  const int n = (int)(quad / ((int64_t)Wq * Ho));
  int iy1, ix1;
  float wy1, wy0;
  bilinear_taps<8>(oy, iy1, wy1, wy0);
  float total[P][CM];
  int lab[2][P];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    f32x4 ta[CM / 4], tb[CM / 4], tc[CM / 4], td[CM / 4];
    {
      float u1, u0;
      bilinear_taps<8>(ox0, ix1, u1, u0);
    }
    head_load_taps<CM>(e == 0 ? Sa : Sb, n, iy1, ix1, Hi, Wi, ta, tb, tc, td);
#pragma unroll
    for (int p = 0; p < P; ++p) {
      int ixp;
      float wx1, wx0;
      bilinear_taps<8>(ox0 + p, ixp, wx1, wx0);
      float sc[CM];
      head_eval_taps<CM>(ta, tb, tc, td, wy1, wy0, wx1, wx0, e == 0 ? ba : bb, C, sc);
      const float m = head_max<CM>(sc, C);
      if (!DIRICHLET) {
        int l = head_label_fast<CM>(sc, m, C);
        if (l < 0) l = head_softmax<CM>(sc, m, C);
        lab[e][p] = l;
      } else {
        head_softmax<CM>(sc, m, C);  // sc = the probabilities the unfused path stores
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < CM; ++k) {
          sc[k] = k < C ? sc[k] : 0.f;
          sum += sc[k];
        }
        {
          const float rs = xv_fast_rcp(sum);
#pragma unroll
          for (int k = 0; k < CM; ++k) sc[k] = k < C ? xv_fast_log(1e-20f + sc[k] * rs) : 0.f;  // renormalise, then log(1e-20 + p)
        }
        // The C dot products sum_k (alpha[c][k] - 1) log p[k], each the fmaf chain over k of dirichlet_fuse_kernel (bit for
        // bit), advanced TWO CLASSES PER INSTRUCTION: v_pk_fma_f32 on the transposed table halves the 2 C^2 = 288 FMAs that
        // made this kernel VALU-bound (padded classes / terms are exact zeros: fma(0, 0, d) = d).
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        f32x2 dot2[CM / 2];
#pragma unroll
        for (int cp = 0; cp < CM / 2; ++cp) dot2[cp] = f32x2{0.f, 0.f};
#pragma unroll
        for (int k = 0; k < CM; ++k) {
          const f32x2* rowk = reinterpret_cast<const f32x2*>(tab + (e * CM + k) * CM);
          const f32x2 lk = f32x2{sc[k], sc[k]};
#pragma unroll
          for (int cp = 0; cp < CM / 2; ++cp) dot2[cp] = __builtin_elementwise_fma(rowk[cp], lk, dot2[cp]);
        }
#pragma unroll
        for (int c = 0; c < CM; ++c) {
          const float L = dot2[c >> 1][c & 1] - ln[e * CM + c];
          total[p][c] = c < C ? (e == 0 ? L : total[p][c] + L) : 0.f;
        }
      }
    }
  }
  int64_t out[P];
#pragma unroll
  for (int p = 0; p < P; ++p) {
    if constexpr (!DIRICHLET) {
      out[p] = dec[lab[0][p] * C + lab[1][p]];
    } else {
      float best = 0.f;
      int bi = 0;
#pragma unroll
      for (int k = 0; k < CM; ++k) {
        const float v = total[p][k] + lp[k];
        if (k < C && (k == 0 || v > best)) {
          best = v;
          bi = k;
        }
      }
      out[p] = bi;
    }
  }
  int64_t* dst = fused + (quad * P);
  if constexpr (P == 4) {
    typedef __attribute__((ext_vector_type(2))) long long i64x2;
    *reinterpret_cast<i64x2*>(dst) = i64x2{out[0], out[1]};
    *reinterpret_cast<i64x2*>(dst + 2) = i64x2{out[2], out[3]};
  } else {
#pragma unroll
    for (int p = 0; p < P; ++p) dst[p] = out[p];
  }
}

// The Dirichlet form of fused_head_kernel for C == CM on PACKED fp32 (v_pk_mul / v_pk_add / v_pk_fma_f32, two classes per
// instruction): every per-class step of the scalar form that is not a summation chain -- the four-tap interpolation, the bias,
// x - max, the products with log2 e / 1 / sum / ln 2, fma(p, 1 / sum', 1e-20), dot - lognorm, + logprior -- is the same IEEE
// operation on the same operands, so the labels stay those of fused_head_kernel<CM, 1, true> and of the unfused path bit for
// bit; the sums (softmax denominator, renormalisation) keep their order.  P consecutive output pixels ox = P m .. P m + P - 1
// per thread share their taps and every table row read from LDS: the 72 16-byte broadcast reads per pixel of the scalar form
// load the LDS pipe about as long as its instructions load the vector ALU.  16 images of 768x384 (tools/dirichlet_head_ab.py):
// scalar 97-105 us; packed, P = 1: 93-100 (VALU instructions 913 -> 751, the LDS reads as before); P = 2: 82; P = 4: 75 us
// (162 VGPRs, 3 waves per SIMD -- the scalar form with four pixels had measured 14 % SLOWER than with one).
// Other class counts, scalar -> this form: 8: 62-69 -> 45-53 us; 16: 135-141 -> 113-121; 20: 185-191 -> 171-178; 24: 254-259 ->
// 312-314; 32: 396-398 -> 576 (328 / 434 registers): the launcher keeps the scalar form above 20 classes.
template <int CM, int P>
__global__ __launch_bounds__(256) void fused_dirichlet_head_pk_kernel(const float* __restrict__ Sa, const float* __restrict__ Sb,
                                                                     const float* __restrict__ ba, const float* __restrict__ bb,
                                                                     int N, int Hi, int Wi, const float* __restrict__ tab_g,
                                                                     const float* __restrict__ lognorm_g,
                                                                     const float* __restrict__ logprior_g,
                                                                     int64_t* __restrict__ fused) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  constexpr int H2 = CM / 2;
  extern __shared__ __attribute__((aligned(16))) float tab[];  // [2][CM k][CM c] = alpha_e[c][k] - 1, lognorm [2][CM], logprior [CM]
  float* ln = tab + 2 * CM * CM;
  float* lp = ln + 2 * CM;
  for (int i = threadIdx.x; i < 2 * CM * CM; i += 256) {
    const int c = i % CM, k = (i / CM) % CM, e = i / (CM * CM);
    tab[i] = tab_g[(e * CM + c) * CM + k];
  }
  for (int i = threadIdx.x; i < 2 * CM; i += 256) ln[i] = lognorm_g[i];
  if (threadIdx.x < CM) lp[threadIdx.x] = logprior_g[threadIdx.x];
  __syncthreads();
  const int Ho = Hi * 8, Wo = Wi * 8, Wq = Wo / P;
  const int64_t nquads = (int64_t)N * Ho * Wq;
  const int64_t quad = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (quad >= nquads) return;
  const int ox0 = (int)(quad % Wq) * P;
  const int oy = (int)((quad / Wq) % Ho);
  const int n = (int)(quad / ((int64_t)Wq * Ho));
  int iy1, ix1;
  float wy1, wy0;
  bilinear_taps<8>(oy, iy1, wy1, wy0);
  {
    float u1, u0;
    bilinear_taps<8>(ox0, ix1, u1, u0);
  }
  f32x2 total[P][H2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    f32x4 ta[CM / 4], tb[CM / 4], tc[CM / 4], td[CM / 4];
    head_load_taps<CM>(e == 0 ? Sa : Sb, n, iy1, ix1, Hi, Wi, ta, tb, tc, td);
    const float* bs = e == 0 ? ba : bb;
    f32x2 lg[P][H2];  // log(1e-20 + p) of the P pixels
#pragma unroll
    for (int p = 0; p < P; ++p) {
      int ixp;
      float wx1, wx0;
      bilinear_taps<8>(ox0 + p, ixp, wx1, wx0);
      const float w00 = wy0 * wx0, w01 = wy0 * wx1, w10 = wy1 * wx0, w11 = wy1 * wx1;
      const f32x2 v00 = f32x2{w00, w00}, v01 = f32x2{w01, w01}, v10 = f32x2{w10, w10}, v11 = f32x2{w11, w11};
      f32x2 s[H2];
#pragma unroll
      for (int j = 0; j < H2; ++j) {  // head_eval_taps: fmaf(d, w11, fmaf(c, w10, fmaf(b, w01, a * w00))) + bias
        const int k4 = j >> 1;
        const f32x2 a2 = (j & 1) ? f32x2{ta[k4].z, ta[k4].w} : f32x2{ta[k4].x, ta[k4].y};
        const f32x2 b2 = (j & 1) ? f32x2{tb[k4].z, tb[k4].w} : f32x2{tb[k4].x, tb[k4].y};
        const f32x2 c2 = (j & 1) ? f32x2{tc[k4].z, tc[k4].w} : f32x2{tc[k4].x, tc[k4].y};
        const f32x2 d2 = (j & 1) ? f32x2{td[k4].z, td[k4].w} : f32x2{td[k4].x, td[k4].y};
        f32x2 t = a2 * v00;
        t = __builtin_elementwise_fma(b2, v01, t);
        t = __builtin_elementwise_fma(c2, v10, t);
        t = __builtin_elementwise_fma(d2, v11, t);
        const f32x2 bias2 = f32x2{bs[2 * j], bs[2 * j + 1]};
        s[j] = t + bias2;
      }
      float m = s[0].x;  // head_max
#pragma unroll
      for (int j = 0; j < H2; ++j) {
        if (j) m = fmaxf(m, s[j].x);
        m = fmaxf(m, s[j].y);
      }
      const f32x2 m2 = f32x2{m, m};
      float sum = 0.f;  // head_softmax: exp(x - max) / sum
#pragma unroll
      for (int j = 0; j < H2; ++j) {
        f32x2 t = s[j] - m2;
        t = t * f32x2{1.4426950408889634f, 1.4426950408889634f};
        s[j] = f32x2{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
        sum += s[j].x;
        sum += s[j].y;
      }
      const float rsum = xv_fast_rcp(sum);
      const f32x2 rsum2 = f32x2{rsum, rsum};
      float sum1 = 0.f;  // the probabilities the unfused path stores, renormalised, log(1e-20 + p)
#pragma unroll
      for (int j = 0; j < H2; ++j) {
        s[j] = s[j] * rsum2;
        sum1 += s[j].x;
        sum1 += s[j].y;
      }
      const float rs = xv_fast_rcp(sum1);
      const f32x2 rs2 = f32x2{rs, rs};
#pragma unroll
      for (int j = 0; j < H2; ++j) {
        const f32x2 t = __builtin_elementwise_fma(s[j], rs2, f32x2{1e-20f, 1e-20f});
        const f32x2 l = f32x2{__builtin_amdgcn_logf(t.x), __builtin_amdgcn_logf(t.y)};
        lg[p][j] = l * f32x2{0.6931471805599453f, 0.6931471805599453f};
      }
    }
    // sum_k (alpha[c][k] - 1) log p[k]: the fmaf chain over k of dirichlet_fuse_kernel, two classes per v_pk_fma_f32
    f32x2 dot[P][H2];
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
      for (int cp = 0; cp < H2; ++cp) dot[p][cp] = f32x2{0.f, 0.f};
#pragma unroll
    for (int k = 0; k < CM; ++k) {
      const f32x2* rowk = reinterpret_cast<const f32x2*>(tab + (e * CM + k) * CM);
#pragma unroll
      for (int cp = 0; cp < H2; ++cp) {
        const f32x2 r = rowk[cp];
#pragma unroll
        for (int p = 0; p < P; ++p) {
          const float lk = lg[p][k >> 1][k & 1];
          dot[p][cp] = __builtin_elementwise_fma(r, f32x2{lk, lk}, dot[p][cp]);
        }
      }
    }
    const f32x2* ln2 = reinterpret_cast<const f32x2*>(ln + e * CM);
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
      for (int cp = 0; cp < H2; ++cp) {
        const f32x2 L = dot[p][cp] - ln2[cp];
        total[p][cp] = e == 0 ? L : total[p][cp] + L;
      }
  }
  const f32x2* lp2 = reinterpret_cast<const f32x2*>(lp);
  int64_t out[P];
#pragma unroll
  for (int p = 0; p < P; ++p) {
    float best = 0.f;
    int bi = 0;
#pragma unroll
    for (int j = 0; j < H2; ++j) {
      const f32x2 v = total[p][j] + lp2[j];
      if (j == 0 || v.x > best) best = v.x, bi = 2 * j;
      if (v.y > best) best = v.y, bi = 2 * j + 1;
    }
    out[p] = bi;
  }
  int64_t* dst = fused + quad * P;
  if constexpr (P % 2 == 0) {
    typedef __attribute__((ext_vector_type(2))) long long i64x2;
#pragma unroll
    for (int p = 0; p < P; p += 2) *reinterpret_cast<i64x2*>(dst + p) = i64x2{out[p], out[p + 1]};
  } else {
#pragma unroll
    for (int p = 0; p < P; ++p) dst[p] = out[p];
  }
}

// ---- softmax + argmax on dense fp32 scores (basic_fusion_model.py:21-22) -------------------------
template <int CMAX>
__global__ __launch_bounds__(256) void softmax_argmax_kernel(const float* __restrict__ score, int64_t npix, int C,
                                                            float* __restrict__ prob, int64_t* __restrict__ label) {
  for (int64_t pix = (int64_t)blockIdx.x * 256 + threadIdx.x; pix < npix; pix += (int64_t)gridDim.x * 256) {
    float sc[CMAX];
    // C == CMAX (a multiple of 4: the 12 classes of the headline model): a pixel's scores / probabilities are whole 16-byte
    // vectors (4-byte accesses at a 48-byte lane stride ran at 0.26 of the HBM rate)
    const bool vec = C == CMAX && (CMAX & 3) == 0 && CMAX != 16 && CMAX != 32;  // (the 16 / 32 forms take unaligned pointers)
    if (vec) {
#pragma unroll
      for (int q = 0; q < CMAX / 4; ++q) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(score + pix * CMAX + 4 * q);
        sc[4 * q] = t.x, sc[4 * q + 1] = t.y, sc[4 * q + 2] = t.z, sc[4 * q + 3] = t.w;
      }
    } else {
#pragma unroll
      for (int k = 0; k < CMAX; ++k) sc[k] = k < C ? score[pix * C + k] : 0.f;
    }
    float m = sc[0];
#pragma unroll
    for (int k = 1; k < CMAX; ++k)
      if (k < C) m = fmaxf(m, sc[k]);
    float e[CMAX];
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < CMAX; ++k) {
      e[k] = k < C ? xv_fast_exp(sc[k] - m) : 0.f;
      sum += e[k];
    }
    const float rsum = xv_fast_rcp(sum);
    float best = -1.f;
    int bi = 0;
#pragma unroll
    for (int k = 0; k < CMAX; ++k) {
      const float p = e[k] * rsum;
      e[k] = p;
      if (k < C) {
        if (prob && !vec) prob[pix * C + k] = p;
        if (p > best) {
          best = p;
          bi = k;
        }
      }
    }
    if (prob && vec) {
#pragma unroll
      for (int q = 0; q < CMAX / 4; ++q)
        *reinterpret_cast<f32x4*>(prob + pix * CMAX + 4 * q) = f32x4{e[4 * q], e[4 * q + 1], e[4 * q + 2], e[4 * q + 3]};
    }
    if (label) label[pix] = bi;
  }
}

// ---- general decoder head: bilinear x8 -> per-channel affine (inference batch norm) -> relu -> 1x1 score ->
// softmax -> argmax, un-commuted.  The default head (score_lowres + decoder_head_kernel) moves the 1x1 conv in
// front of the interpolation, which is only valid while relu(up8(f)) == up8(f); a batch norm with a non-zero shift
// between the deconv and its relu (custom_layers.py:112-119, the default of decoder() when fusion_fcn.py:38
// calls it) breaks that, and this kernel interpolates all U features per pixel instead (16x the FMAs).
// One thread = 4 horizontally consecutive output pixels sharing the same 2x2 source pixels.
template <int CM, bool CLAMP>
__device__ inline void head_affine_group(const u32x4& a00, const u32x4& a01, const u32x4& a10, const u32x4& a11, float wy0,
                                  float wy1, const float (&wx0)[4], const float (&wx1)[4],
                                  const float* __restrict__ wrow, const float* __restrict__ srow,
                                  const float* __restrict__ trow, int C, int remain, float (&sc)[4][CM]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int sh = (i & 1) * 16, q = i >> 1;
    const float f00 = bf16_bits_to_f32((a00[q] >> sh) & 0xffffu), f01 = bf16_bits_to_f32((a01[q] >> sh) & 0xffffu);
    const float f10 = bf16_bits_to_f32((a10[q] >> sh) & 0xffffu), f11 = bf16_bits_to_f32((a11[q] >> sh) & 0xffffu);
    const float v0 = f00 * wy0 + f10 * wy1;  // source column ix0
    const float v1 = f01 * wy0 + f11 * wy1;  // source column ix1
    float up[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) up[j] = fmaxf((v0 * wx0[j] + v1 * wx1[j]) * srow[i] + trow[i], 0.f);
    if (CLAMP && i * C + CM > remain) {  // wave-uniform
#pragma unroll
      for (int k = 0; k < CM; ++k) {
        int off = i * C + k;
        off = off < remain ? off : remain - 1;
        const float wv = wrow[off];
#pragma unroll
        for (int j = 0; j < 4; ++j) sc[j][k] = fmaf(up[j], wv, sc[j][k]);
      }
    } else {
#pragma unroll
      for (int k = 0; k < CM; ++k) {
        const float wv = wrow[i * C + k];
#pragma unroll
        for (int j = 0; j < 4; ++j) sc[j][k] = fmaf(up[j], wv, sc[j][k]);
      }
    }
  }
}

template <int CM>
__global__ __launch_bounds__(256) void decoder_head_affine_kernel(const __bf16* __restrict__ f, const float* __restrict__ sc_g,
                                                                 const float* __restrict__ sh_g, const float* __restrict__ ws_g,
                                                          const float* __restrict__ bs_g, int N, int Hi, int Wi, int U,
                                                          int C, float* __restrict__ score, float* __restrict__ prob,
                                                          int64_t* __restrict__ label) {
  const int Ho = Hi * 8, Wo = Wi * 8;
  // block = 128 x 8 output pixels: a wave covers 2 rows x 128 columns
  const int tilesx = (Wo + 127) / 128;
  const int tx = blockIdx.x % tilesx;
  int r = blockIdx.x / tilesx;
  const int tilesy = Hi;  // Ho / 8
  const int ty = r % tilesy;
  const int n = r / tilesy;
  const int ox = tx * 128 + (threadIdx.x & 31) * 4, oy = ty * 8 + (threadIdx.x >> 5);
  if (ox >= Wo) return;
  int iy1, ix1;
  float wy1, wy0, wx1[4], wx0[4];
  bilinear_taps<8>(oy, iy1, wy1, wy0);
#pragma unroll
  for (int j = 0; j < 4; ++j) bilinear_taps<8>(ox + j, ix1, wx1[j], wx0[j]);  // same ix1 for the 4 aligned pixels
  const __bf16* p00 = f + (((int64_t)n * (Hi + 2) + iy1) * (Wi + 2) + ix1) * U;
  const int64_t rowp = (int64_t)(Wi + 2) * U;
  float sc[4][CM];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int k = 0; k < CM; ++k) sc[j][k] = 0.f;
  // software pipeline: the four 16-byte source loads of channel group g+1 are in flight while group g is
  // multiplied (few waves per SIMD fit beside 4*CM accumulators, so the loop must hide its own L2 latency)
  u32x4 n00 = *reinterpret_cast<const u32x4*>(p00), n01 = *reinterpret_cast<const u32x4*>(p00 + U);
  u32x4 n10 = *reinterpret_cast<const u32x4*>(p00 + rowp), n11 = *reinterpret_cast<const u32x4*>(p00 + rowp + U);
  for (int u0 = 0; u0 < U; u0 += 8) {
    const u32x4 a00 = n00, a01 = n01, a10 = n10, a11 = n11;
    const int un = u0 + 8 < U ? u0 + 8 : u0;
    n00 = *reinterpret_cast<const u32x4*>(p00 + un);
    n01 = *reinterpret_cast<const u32x4*>(p00 + U + un);
    n10 = *reinterpret_cast<const u32x4*>(p00 + rowp + un);
    n11 = *reinterpret_cast<const u32x4*>(p00 + rowp + U + un);
    const float* wrow = ws_g + u0 * C;
    if (u0 + 16 <= U)
      head_affine_group<CM, false>(a00, a01, a10, a11, wy0, wy1, wx0, wx1, wrow, sc_g + u0, sh_g + u0, C, 0, sc);
    else
      head_affine_group<CM, true>(a00, a01, a10, a11, wy0, wy1, wx0, wx1, wrow, sc_g + u0, sh_g + u0, C, (U - u0) * C, sc);
  }
  const int64_t opix = ((int64_t)n * Ho + oy) * Wo + ox;
  int64_t lab[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
#pragma unroll
    for (int k = 0; k < CM; ++k) sc[j][k] += bs_g[k < C ? k : C - 1];
    if (score) {
#pragma unroll
      for (int k = 0; k < CM; ++k)
        if (k < C) score[(opix + j) * C + k] = sc[j][k];
    }
    float m = sc[j][0];
#pragma unroll
    for (int k = 1; k < CM; ++k)
      if (k < C) m = fmaxf(m, sc[j][k]);
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < CM; ++k) {
      sc[j][k] = k < C ? xv_fast_exp(sc[j][k] - m) : 0.f;
      sum += sc[j][k];
    }
    const float rsum = xv_fast_rcp(sum);
    float best = -1.f;
    int bi = 0;
#pragma unroll
    for (int k = 0; k < CM; ++k) {
      const float p = sc[j][k] * rsum;
      if (k < C) {
        if (prob) prob[(opix + j) * C + k] = p;
        if (p > best) {
          best = p;
          bi = k;
        }
      }
    }
    lab[j] = bi;
  }
  if (label) {
    typedef __attribute__((ext_vector_type(2))) int64_t i64x2;
    *reinterpret_cast<i64x2*>(label + opix) = i64x2{lab[0], lab[1]};
    *reinterpret_cast<i64x2*>(label + opix + 2) = i64x2{lab[2], lab[3]};
  }
}


__global__ __launch_bounds__(256) void concat_kernel(const u32x4* __restrict__ a, const u32x4* __restrict__ b,
                                                    u32x4* __restrict__ y, int64_t pixels, int ca8, int cb8) {
  const int cy8 = ca8 + cb8;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < pixels * cy8; idx += (int64_t)gridDim.x * 256) {
    const int64_t p = idx / cy8;
    const int c = (int)(idx - p * cy8);
    y[idx] = c < ca8 ? a[p * ca8 + c] : b[p * cb8 + (c - ca8)];
  }
}

inline int grid_for(int64_t total, int per_block = 256, int cap = 8192) {
  int64_t g = (total + per_block - 1) / per_block;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

// block_0_1 + the gather of block_0_2 in one pass (conv_first_mfma_kernel<CIN, false, true>): z as xv_gather_conv7s2 writes it.
extern "C" int xv_conv2d_first_gather7s2_fwd(const float* x, int n, int h, int w, int cin, const float* w_hwio,
                                             const float* bias, const xv_act* z, int relu, void* stream) {
  XV_CHECK_ARG(x && w_hwio && bias && z && z->data);
  XV_CHECK_ARG(z->dtype == XV_BF16);
  XV_CHECK_SHAPE(n > 0 && h > 0 && w >= 16 && (cin == 1 || cin == 3) && (w & 15) == 0 && (h & 1) == 0 &&
                 (int64_t)n * h * w * cin < 0x7ff00000);
  XV_CHECK_SHAPE(z->n == n && z->h == h / 2 && z->w == w / 2 && z->c == 576);
  static const int per_cu = getenv("XV_FIRST_WG_PER_CU") ? atoi(getenv("XV_FIRST_WG_PER_CU")) : 4;
  const int64_t ntiles = (int64_t)n * h * (w / 16);
  const int64_t want = (ntiles + 3) / 4, cap = (int64_t)xv_num_cus() * (per_cu > 0 ? per_cu : 4);
  const int64_t g0 = want < cap ? want : cap;
  const int tpw = (int)((ntiles + g0 * 4 - 1) / (g0 * 4));
  const unsigned g2 = (unsigned)((ntiles + (int64_t)tpw * 4 - 1) / ((int64_t)tpw * 4));
  hipStream_t s = (hipStream_t)stream;
  if (cin == 1)
    hipLaunchKernelGGL((conv_first_mfma_kernel<1, false, true>), dim3(g2), dim3(256), 0, s, x, w_hwio, bias, (__bf16*)z->data, n, h,
                       w, relu, tpw, 1.f);
  else
    hipLaunchKernelGGL((conv_first_mfma_kernel<3, false, true>), dim3(g2), dim3(256), 0, s, x, w_hwio, bias, (__bf16*)z->data, n, h,
                       w, relu, tpw, 1.f);
  return xv_launch_status();
}

extern "C" int xv_conv2d_first_fwd(const float* x, int n, int h, int w, int cin, const float* w_hwio,
                                   const float* bias, const xv_act* y, int relu, void* stream) {
  XV_CHECK_ARG(x && w_hwio && bias && y && y->data);
  XV_CHECK_ARG(y->dtype == XV_BF16 || y->dtype == XV_FP8);
  XV_CHECK_SHAPE(n > 0 && h > 0 && w > 0 && cin >= 1 && cin <= 4);
  XV_CHECK_SHAPE(y->n == n && y->h == h && y->w == w && y->c == 64);
  const bool out8 = y->dtype == XV_FP8;  // e4m3 output: the matrix-core kernel only (1 / 3 channels, w % 16 == 0)
  if (out8) XV_CHECK_SHAPE((cin == 1 || cin == 3) && (w & 15) == 0 && y->scale_exp > -100 && y->scale_exp < 100);
  const float out_mul = out8 ? exp2f((float)-y->scale_exp) : 1.f;
  XV_CHECK_SHAPE((w & 1) == 0 && w >= 16 && (int64_t)n * h * (w / 2) < 0x7fffff00);
  const int64_t npair = (int64_t)n * h * (w / 2);
  const unsigned grid = (unsigned)((npair + 255) / 256);
  hipStream_t s = (hipStream_t)stream;
  __bf16* yp = (__bf16*)y->data;
  // The MFMA form takes whole 16-pixel tiles and 32-bit float offsets into x; XV_FIRST_OLD=1 keeps the FMA kernel
  // (A/B timing), XV_FIRST_WG_PER_CU sizes the persistent grid.
  static const bool use_old = getenv("XV_FIRST_OLD") != nullptr;
  if ((cin == 1 || cin == 3) && (w & 15) == 0 && (int64_t)n * h * w * cin < 0x7ff00000 && (!use_old || out8)) {
    // 98 VGPRs: five workgroups resident per CU; measured at 8 x 384 x 768 with grid-stride tiles: 5 per CU (one round)
    // 78 / 102 us (depth / RGB), 8: 70 / 92, 16: 65 / 92, 32: 67 / 98, 64: 83 / 122; with contiguous runs per wave 8 per CU:
    // 59 / 84, 16: 62 / 96, 32: 71 / 101; 4 or 5 per CU (all resident, one round): 57 / 83; 3: 88 / 100; 6 (one more than
    // fits): 71 / 97.  4: still one round if a rebuild needs a few more registers
    static const int per_cu = getenv("XV_FIRST_WG_PER_CU") ? atoi(getenv("XV_FIRST_WG_PER_CU")) : 4;
    const int64_t ntiles = (int64_t)n * h * (w / 16);
    const int64_t want = (ntiles + 3) / 4, cap = (int64_t)xv_num_cus() * (per_cu > 0 ? per_cu : 4);
    const int64_t g0 = want < cap ? want : cap;
    const int tpw = (int)((ntiles + g0 * 4 - 1) / (g0 * 4));               // tiles per wave
    const unsigned g2 = (unsigned)((ntiles + (int64_t)tpw * 4 - 1) / ((int64_t)tpw * 4));
    if (out8) {
      if (cin == 1)
        hipLaunchKernelGGL((conv_first_mfma_kernel<1, true>), dim3(g2), dim3(256), 0, s, x, w_hwio, bias, yp, n, h, w, relu, tpw,
                           out_mul);
      else
        hipLaunchKernelGGL((conv_first_mfma_kernel<3, true>), dim3(g2), dim3(256), 0, s, x, w_hwio, bias, yp, n, h, w, relu, tpw,
                           out_mul);
    } else if (cin == 1)
      hipLaunchKernelGGL(conv_first_mfma_kernel<1>, dim3(g2), dim3(256), 0, s, x, w_hwio, bias, yp, n, h, w, relu, tpw, 1.f);
    else
      hipLaunchKernelGGL(conv_first_mfma_kernel<3>, dim3(g2), dim3(256), 0, s, x, w_hwio, bias, yp, n, h, w, relu, tpw, 1.f);
    return xv_launch_status();
  }
  if (out8) return XV_ESHAPE;
  switch (cin) {
    case 1: hipLaunchKernelGGL(conv_first_kernel<1>, dim3(grid), dim3(256), 0, s, x, w_hwio, bias, yp, n, h, w, relu); break;
    case 2: hipLaunchKernelGGL(conv_first_kernel<2>, dim3(grid), dim3(256), 0, s, x, w_hwio, bias, yp, n, h, w, relu); break;
    case 3: hipLaunchKernelGGL(conv_first_kernel<3>, dim3(grid), dim3(256), 0, s, x, w_hwio, bias, yp, n, h, w, relu); break;
    default: hipLaunchKernelGGL(conv_first_kernel<4>, dim3(grid), dim3(256), 0, s, x, w_hwio, bias, yp, n, h, w, relu); break;
  }
  return xv_launch_status();
}

extern "C" int xv_maxpool2x2_fwd(const xv_act* x, const xv_act* y, void* stream) {
  XV_REQUIRE_BF16(x, y);
  XV_CHECK_ARG(x && y && x->data && y->data);
  XV_CHECK_SHAPE(x->n > 0 && x->h > 0 && x->w > 0 && x->c > 0 && (x->c & 7) == 0 && (x->h & 1) == 0 && (x->w & 1) == 0);
  XV_CHECK_SHAPE(y->n == x->n && y->h == x->h / 2 && y->w == x->w / 2 && y->c == x->c);
  const int64_t total = (int64_t)y->n * y->h * y->w * (y->c >> 3);
  hipLaunchKernelGGL(maxpool_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const __bf16*)x->data,
                     (__bf16*)y->data, y->n, y->h, y->w, y->c);
  return xv_launch_status();
}

extern "C" int xv_upsample2x_affine_act_add(const xv_act* x, const float* scale, const float* shift,
                                           const xv_act* residual, const xv_act* y, int relu, void* stream) {
  XV_REQUIRE_BF16(x, residual, y);
  XV_CHECK_ARG(x && y && x->data && y->data);
  XV_CHECK_ARG((scale == nullptr) == (shift == nullptr));
  XV_CHECK_SHAPE(x->n > 0 && x->h > 0 && x->w > 0 && x->c > 0 && (x->c & 7) == 0);
  XV_CHECK_SHAPE(y->n == x->n && y->h == 2 * x->h && y->w == 2 * x->w && y->c == x->c);
  const __bf16* res = nullptr;
  if (residual && residual->data) {
    XV_CHECK_SHAPE(residual->n == y->n && residual->h == y->h && residual->w == y->w && residual->c == y->c);
    res = (const __bf16*)residual->data;
  }
  const int64_t total = (int64_t)y->n * y->h * y->w * (y->c >> 3);
  hipLaunchKernelGGL(upsample2x_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                     (const __bf16*)x->data, res, scale, shift, (__bf16*)y->data, x->n, x->h, x->w, x->c, relu);
  return xv_launch_status();
}

extern "C" int xv_upsample2x_affine_relu_add(const xv_act* x, const float* scale, const float* shift,
                                            const xv_act* residual, const xv_act* y, void* stream) {
  XV_REQUIRE_BF16(x, residual, y);
  return xv_upsample2x_affine_act_add(x, scale, shift, residual, y, 1, stream);
}

extern "C" int xv_upsample2x_relu_add(const xv_act* x, const xv_act* residual, const xv_act* y, void* stream) {
  XV_REQUIRE_BF16(x, residual, y);
  return xv_upsample2x_affine_relu_add(x, nullptr, nullptr, residual, y, stream);
}

// Depth-to-space behind the dense transposed convolution (xv_deconv_dense_fwd): z holds the s*s output phases of every
// input pixel side by side in its channels ([(py*s + px)*C + c]); y[n][qy*s + py][qx*s + px][c] = act(z * scale + shift)
// [+ residual].  One thread = 8 channels of one output pixel; for a fixed output row the s*C channels of an input pixel
// are contiguous, so reads and writes are both 16-byte coalesced.
__global__ __launch_bounds__(256) void depth_to_space_kernel(const __bf16* __restrict__ z, const __bf16* __restrict__ res,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            __bf16* __restrict__ y, int N, int Hi, int Wi, int C, int S,
                                                            int relu) {
  const int c8 = C >> 3;
  const int Ho = Hi * S, Wo = Wi * S;
  const int64_t total = (int64_t)N * Ho * Wo * c8;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int cg = (int)(idx % c8);
    int64_t r = idx / c8;
    const int ox = (int)(r % Wo);
    r /= Wo;
    const int oy = (int)(r % Ho);
    const int n = (int)(r / Ho);
    const int qy = oy / S, py = oy - qy * S, qx = ox / S, px = ox - qx * S;
    const u32x4 a = *reinterpret_cast<const u32x4*>(z + (((int64_t)n * (Hi + 2) + qy + 1) * (Wi + 2) + qx + 1) * ((int64_t)S * S * C) +
                                                    (int64_t)(py * S + px) * C + cg * 8);
    const int64_t yoff = (((int64_t)n * (Ho + 2) + oy + 1) * (Wo + 2) + ox + 1) * C + cg * 8;
    u32x4 rr = {0, 0, 0, 0};
    if (res != nullptr) rr = *reinterpret_cast<const u32x4*>(res + yoff);
    u32x4 o;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      float lo = bf16_bits_to_f32(a[w] & 0xffffu), hi = __builtin_bit_cast(float, a[w] & 0xffff0000u);
      if (scale != nullptr) {
        lo = lo * scale[cg * 8 + 2 * w] + shift[cg * 8 + 2 * w];
        hi = hi * scale[cg * 8 + 2 * w + 1] + shift[cg * 8 + 2 * w + 1];
      }
      if (relu) {
        lo = fmaxf(lo, 0.f);
        hi = fmaxf(hi, 0.f);
      }
      lo += bf16_bits_to_f32(rr[w] & 0xffffu);
      hi += __builtin_bit_cast(float, rr[w] & 0xffff0000u);
      o[w] = pack_bf16x2(lo, hi);
    }
    *reinterpret_cast<u32x4*>(y + yoff) = o;
  }
}

int xv_launch_depth_to_space(const xv_act* z, const float* scale, const float* shift, const xv_act* residual, const xv_act* y,
                             int stride, int relu, hipStream_t stream) {
  const int64_t total = (int64_t)y->n * y->h * y->w * (y->c >> 3);
  hipLaunchKernelGGL(depth_to_space_kernel, dim3(grid_for(total)), dim3(256), 0, stream, (const __bf16*)z->data,
                     residual && residual->data ? (const __bf16*)residual->data : nullptr, scale, shift, (__bf16*)y->data, z->n,
                     z->h, z->w, y->c, stride, relu);
  return xv_launch_status();
}

// tf.layers.dropout(x, rate, training=True) (simple_fcn.py:50-62,71-78,124-126: the MC-dropout sites of encoder /
// decoder): each element is kept with probability 1 - rate and scaled by 1 / (1 - rate), else zero.  Counter-based
// random bits: a 64-bit mix of (seed, element index) per element, so a mask depends only on the seed, not on the launch
// geometry.  Runs over the whole padded buffer (the zero border stays zero).  One thread = 8 channels.
__device__ __forceinline__ uint32_t xv_mix32(uint64_t z) {  // splitmix64 finaliser, upper 32 bits
  z += 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return (uint32_t)((z ^ (z >> 31)) >> 32);
}

__global__ __launch_bounds__(256) void dropout_kernel(const u32x4* __restrict__ x, u32x4* __restrict__ y, int64_t total8,
                                                     uint32_t drop_below, float scale, uint64_t seed) {
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total8; idx += (int64_t)gridDim.x * 256) {
    const u32x4 v = x[idx];
    u32x4 o;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const uint32_t r0 = xv_mix32(seed ^ (uint64_t)(idx * 8 + 2 * w) * 0xd1342543de82ef95ull);
      const uint32_t r1 = xv_mix32(seed ^ (uint64_t)(idx * 8 + 2 * w + 1) * 0xd1342543de82ef95ull);
      const float lo = r0 >= drop_below ? bf16_bits_to_f32(v[w] & 0xffffu) * scale : 0.f;
      const float hi = r1 >= drop_below ? __builtin_bit_cast(float, v[w] & 0xffff0000u) * scale : 0.f;
      o[w] = pack_bf16x2(lo, hi);
    }
    y[idx] = o;
  }
}

extern "C" int xv_dropout(const xv_act* x, const xv_act* y, float rate, uint64_t seed, void* stream) {
  XV_REQUIRE_BF16(x, y);
  XV_CHECK_ARG(x && y && x->data && y->data && rate >= 0.f && rate < 1.f);
  XV_CHECK_SHAPE(x->n == y->n && x->h == y->h && x->w == y->w && x->c == y->c && (x->c & 7) == 0);
  XV_CHECK_SHAPE(x->dtype == XV_BF16 && y->dtype == XV_BF16);
  const int64_t total8 = (int64_t)x->n * (x->h + 2) * (x->w + 2) * (x->c >> 3);
  const double thr = (double)rate * 4294967296.0;
  hipLaunchKernelGGL(dropout_kernel, dim3(grid_for(total8)), dim3(256), 0, (hipStream_t)stream, (const u32x4*)x->data,
                     (u32x4*)y->data, total8, (uint32_t)(thr > 4294967295.0 ? 4294967295.0 : thr), 1.f / (1.f - rate), seed);
  return xv_launch_status();
}

// y[..., :Ca] = a, y[..., Ca:] = b over the whole padded buffers (tf.concat(axis=3), fusion_fcn.py:27-28)
extern "C" int xv_concat_channels(const xv_act* a, const xv_act* b, const xv_act* y, void* stream) {
  XV_REQUIRE_BF16(a, b, y);
  XV_CHECK_ARG(a && b && y && a->data && b->data && y->data);
  XV_CHECK_SHAPE(a->n == b->n && a->h == b->h && a->w == b->w && y->n == a->n && y->h == a->h && y->w == a->w);
  XV_CHECK_SHAPE(y->c == a->c + b->c && (a->c & 7) == 0 && (b->c & 7) == 0);
  const int64_t total = (int64_t)y->n * (y->h + 2) * (y->w + 2) * (y->c >> 3);
  hipLaunchKernelGGL(concat_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const u32x4*)a->data,
                     (const u32x4*)b->data, (u32x4*)y->data, (int64_t)y->n * (y->h + 2) * (y->w + 2), a->c >> 3,
                     b->c >> 3);
  return xv_launch_status();
}

// S = fused . Ws at 1/8 resolution into a zero-bordered fp32 [N][h+2][w+2][CM] buffer (shared by the
// forward head and the head backward)
extern "C" int xv_score_lowres(const xv_act* fused, const float* w_score, int num_classes, float* S, void* stream) {
  XV_REQUIRE_BF16(fused);
  XV_CHECK_ARG(fused && fused->data && w_score && S);
  XV_CHECK_SHAPE(fused->c > 0 && (fused->c & 7) == 0 && num_classes >= 1 && num_classes <= 32);
  const int64_t lowres = (int64_t)fused->n * (fused->h + 2) * (fused->w + 2);
  const unsigned g1 = (unsigned)((lowres + 127) / 128);
  hipStream_t s = (hipStream_t)stream;
  const __bf16* f = (const __bf16*)fused->data;
  XV_CHECK_SHAPE(fused->c <= 256);
#define XV_SL(CMV)                                                                                             \
  hipLaunchKernelGGL(score_lowres_kernel<CMV>, dim3(g1), dim3(128), (size_t)fused->c * CMV * 4, s, f, w_score, \
                     fused->n, fused->h, fused->w, fused->c, num_classes, S)
  switch ((num_classes + 3) / 4) {
    case 1: XV_SL(4); break;
    case 2: XV_SL(8); break;
    case 3: XV_SL(12); break;
    case 4: XV_SL(16); break;
    case 5: XV_SL(20); break;
    case 6: XV_SL(24); break;
    case 7: XV_SL(28); break;
    default: XV_SL(32); break;
  }
#undef XV_SL
  return xv_launch_status();
}

// Fused head of a two-expert fusion model (see fused_head_kernel): Sa / Sb from xv_score_lowres of each expert's `fused`
// map; mode 0 = Bayes (tab = loglik [2][C][C], lognorm unused), 1 = Dirichlet (tab = am1 [2][C][C], lognorm [2][C]).
extern "C" int xv_fused_head_fwd(const float* Sa, const float* Sb, const float* bias_a, const float* bias_b, int n, int hi,
                                 int wi, int num_classes, int mode, const float* tab, const float* lognorm,
                                 const float* logprior, int64_t* fused_label, void* stream) {
  XV_CHECK_ARG(Sa && Sb && bias_a && bias_b && tab && logprior && fused_label && (mode == 0 || (mode == 1 && lognorm)));
  XV_CHECK_SHAPE(n > 0 && hi > 0 && wi > 0 && num_classes >= 1 && num_classes <= 32);
  // threads: four output pixels each (Bayes; Dirichlet in the packed form), one (Dirichlet, scalar form) -- see fused_head_kernel
  const int64_t nthreads = (int64_t)n * hi * 8 * wi * (mode == 0 ? 2 : 8);
  XV_CHECK_ARG(((uintptr_t)fused_label & 15) == 0);
  const unsigned grid = (unsigned)((nthreads + 255) / 256);
  hipStream_t s = (hipStream_t)stream;
  // XV_DIRICHLET_HEAD_PK=0: the scalar form of the Dirichlet head for C == CM too (read per call: the test that pins the two
  // forms to the same labels switches it)
  const char* pk_env = getenv("XV_DIRICHLET_HEAD_PK");
  const bool pk = !(pk_env && pk_env[0] == '0');
#define XV_FH(CMV)                                                                                                      \
  {                                                                                                                     \
    const size_t lds = (size_t)(2 * (mode == 0 ? num_classes : CMV) * CMV + 3 * CMV + (mode == 0 ? num_classes * num_classes : 0)) * 4; \
    if (mode == 0 && num_classes == CMV)                                                                                \
      hipLaunchKernelGGL((fused_head_kernel<CMV, 0, true>), dim3(grid), dim3(256), lds, s, Sa, Sb, bias_a, bias_b, n, hi, \
                         wi, num_classes, tab, lognorm, logprior, fused_label);                                         \
    else if (mode == 0)                                                                                                 \
      hipLaunchKernelGGL((fused_head_kernel<CMV, 0>), dim3(grid), dim3(256), lds, s, Sa, Sb, bias_a, bias_b, n, hi, wi,  \
                         num_classes, tab, lognorm, logprior, fused_label);                                             \
    else if (num_classes == CMV && pk && CMV <= 20) /* (the template argument below only keeps CMV > 20 from instantiating) */ \
      hipLaunchKernelGGL((fused_dirichlet_head_pk_kernel<(CMV <= 20 ? CMV : 4), 4>), dim3((grid + 3) / 4), dim3(256), lds, s, \
                         Sa, Sb, bias_a, bias_b, n, hi, wi, tab, lognorm, logprior, fused_label);                       \
    else if (num_classes == CMV)                                                                                        \
      hipLaunchKernelGGL((fused_head_kernel<CMV, 1, true>), dim3(grid), dim3(256), lds, s, Sa, Sb, bias_a, bias_b, n, hi, \
                         wi, num_classes, tab, lognorm, logprior, fused_label);                                         \
    else                                                                                                                \
      hipLaunchKernelGGL((fused_head_kernel<CMV, 1>), dim3(grid), dim3(256), lds, s, Sa, Sb, bias_a, bias_b, n, hi, wi,  \
                         num_classes, tab, lognorm, logprior, fused_label);                                             \
  }
  switch ((num_classes + 3) / 4) {
    case 1: XV_FH(4); break;
    case 2: XV_FH(8); break;
    case 3: XV_FH(12); break;
    case 4: XV_FH(16); break;
    case 5: XV_FH(20); break;
    case 6: XV_FH(24); break;
    case 7: XV_FH(28); break;
    default: XV_FH(32); break;
  }
#undef XV_FH
  return xv_launch_status();
}

extern "C" int xv_decoder_head_affine_fwd(const xv_act* fused, const float* scale, const float* shift,
                                         const float* w_score, const float* b_score, int num_classes, float* score,
                                         float* prob, int64_t* label, void* stream) {
  XV_REQUIRE_BF16(fused);
  XV_CHECK_ARG(fused && fused->data && scale && shift && w_score && b_score && (score || prob || label));
  XV_CHECK_SHAPE(fused->c > 0 && (fused->c & 7) == 0 && num_classes >= 1 && num_classes <= 32);
  const int Wo = fused->w * 8;
  const int64_t nblk = (int64_t)((Wo + 127) / 128) * fused->h * fused->n;
  XV_CHECK_SHAPE(nblk <= 0x7fffffff);
  hipStream_t s = (hipStream_t)stream;
  const __bf16* f = (const __bf16*)fused->data;
#define XV_HA(CMV)                                                                                                  \
  hipLaunchKernelGGL(decoder_head_affine_kernel<CMV>, dim3((unsigned)nblk), dim3(256), 0, s, f, scale, shift, w_score, \
                     b_score, fused->n, fused->h, fused->w, fused->c, num_classes, score, prob, label)
  switch ((num_classes + 3) / 4) {
    case 1: XV_HA(4); break;
    case 2: XV_HA(8); break;
    case 3: XV_HA(12); break;
    case 4: XV_HA(16); break;
    case 5: XV_HA(20); break;
    case 6: XV_HA(24); break;
    case 7: XV_HA(28); break;
    default: XV_HA(32); break;
  }
#undef XV_HA
  return xv_launch_status();
}

extern "C" size_t xv_decoder_head_workspace_bytes(int n, int h, int w, int num_classes) {
  if (!xv_dims_sane(n, h, w) || num_classes < 1 || num_classes > 32) return 0;
  return (size_t)n * ((size_t)h + 2) * ((size_t)w + 2) * ((num_classes + 3) / 4 * 4) * sizeof(float);
}

// low-resolution class scores -> score / prob / label: the label alone (16-byte aligned) through the four-pixel form
static void launch_decoder_head(const float* S, const float* b_score, int n, int hi, int wi, int num_classes, float* score,
                                float* prob, int64_t* label, hipStream_t s) {
  const int64_t npix = (int64_t)n * hi * wi * 64;
  const bool label_only = !score && !prob && ((uintptr_t)label & 15) == 0;
  const unsigned g2 = (unsigned)(((label_only ? npix / 4 : npix) + 255) / 256);
#define XV_HEAD(CMV)                                                                                                     \
  {                                                                                                                      \
    if (label_only)                                                                                                      \
      hipLaunchKernelGGL(decoder_head_label4_kernel<CMV>, dim3(g2), dim3(256), 0, s, S, b_score, n, hi, wi, num_classes, \
                         label);                                                                                         \
    else                                                                                                                 \
      hipLaunchKernelGGL(decoder_head_kernel<CMV>, dim3(g2), dim3(256), 0, s, S, b_score, n, hi, wi, num_classes, score, \
                         prob, label);                                                                                   \
  }
  switch ((num_classes + 3) / 4) {
    case 1: XV_HEAD(4); break;
    case 2: XV_HEAD(8); break;
    case 3: XV_HEAD(12); break;
    case 4: XV_HEAD(16); break;
    case 5: XV_HEAD(20); break;
    case 6: XV_HEAD(24); break;
    case 7: XV_HEAD(28); break;
    default: XV_HEAD(32); break;
  }
#undef XV_HEAD
}

extern "C" int xv_decoder_head_fwd(const xv_act* fused, const float* w_score, const float* b_score, int num_classes,
                                   float* score, float* prob, int64_t* label, void* workspace, size_t workspace_bytes,
                                   void* stream) {
  XV_REQUIRE_BF16(fused);
  XV_CHECK_ARG(fused && fused->data && w_score && b_score && workspace);
  XV_CHECK_ARG(score || prob || label);
  XV_CHECK_SHAPE(fused->n > 0 && fused->h > 0 && fused->w > 0);
  XV_CHECK_SHAPE(fused->c > 0 && (fused->c & 7) == 0 && num_classes >= 1 && num_classes <= 32);
  if (workspace_bytes < xv_decoder_head_workspace_bytes(fused->n, fused->h, fused->w, num_classes)) return XV_EWORKSPACE;
  XV_CHECK_ARG(((uintptr_t)workspace & 15) == 0);
  hipStream_t s = (hipStream_t)stream;
  float* S = (float*)workspace;
  const int64_t npix = (int64_t)fused->n * fused->h * fused->w * 64;
  XV_CHECK_SHAPE((npix + 255) / 256 <= 0x7fffffff);
  {
    const int rc = xv_score_lowres(fused, w_score, num_classes, S, stream);
    if (rc != XV_OK) return rc;
  }
  launch_decoder_head(S, b_score, fused->n, fused->h, fused->w, num_classes, score, prob, label, s);
  return xv_launch_status();
}

// The second half of xv_decoder_head_fwd alone: low-resolution class scores S (float32 [N][hi+2][wi+2][CP], zero border;
// from xv_score_lowres or xv_score_lowres_f32) -> score / prob / label at 8x the resolution.
extern "C" int xv_decoder_head_from_scores(const float* S, const float* b_score, int n, int hi, int wi, int num_classes,
                                           float* score, float* prob, int64_t* label, void* stream) {
  XV_CHECK_ARG(S && b_score && (score || prob || label));
  XV_CHECK_SHAPE(n > 0 && hi > 0 && wi > 0 && num_classes >= 1 && num_classes <= 32);
  const int64_t npix = (int64_t)n * hi * wi * 64;
  XV_CHECK_SHAPE((npix + 255) / 256 <= 0x7fffffff);
  launch_decoder_head(S, b_score, n, hi, wi, num_classes, score, prob, label, (hipStream_t)stream);
  return xv_launch_status();
}

extern "C" int xv_softmax_argmax(const float* score, int64_t npix, int num_classes, float* prob, int64_t* label,
                                 void* stream) {
  XV_CHECK_ARG(score && (prob || label));
  XV_CHECK_SHAPE(npix > 0 && num_classes >= 1 && num_classes <= 32);
  hipStream_t s = (hipStream_t)stream;
  const bool al16 = (((uintptr_t)score | (uintptr_t)prob) & 15) == 0;  // (the vector form of a 12-class map: xv_softmax_argmax
  //                                                                         takes any float pointer)
  if (num_classes == 12 && al16)
    hipLaunchKernelGGL(softmax_argmax_kernel<12>, dim3(grid_for(npix)), dim3(256), 0, s, score, npix, num_classes, prob,
                       label);
  else if (num_classes <= 16)
    hipLaunchKernelGGL(softmax_argmax_kernel<16>, dim3(grid_for(npix)), dim3(256), 0, s, score, npix, num_classes, prob,
                       label);
  else
    hipLaunchKernelGGL(softmax_argmax_kernel<32>, dim3(grid_for(npix)), dim3(256), 0, s, score, npix, num_classes, prob,
                       label);
  return xv_launch_status();
}
