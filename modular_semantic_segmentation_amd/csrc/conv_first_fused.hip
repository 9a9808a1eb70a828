// conv1_1 + conv1_2 (+ pool1) in ONE kernel: the first two layers of the FCN trunk (xview/models/simple_fcn.py:39-41:
// conv2d(3x3, 64, relu) on the raw input, conv2d(3x3, 64, relu), max_pooling2d) -- configuration-free entry
// xv_conv_first_pair_fwd, inference only (training keeps conv1_1's output for the filter gradient).
//
// Why.  conv1_1 writes its 64-channel map (609 MB per expert at 16 images of 768x384) and conv1_2 reads it back; both ends
// are bound by what one CU can move -- conv1_1 by the vector store path (~16 B/clk/CU: 176 us RGB / 163 us depth, 0.47 of the
// HBM rate), conv1_2 by the LDS-DMA path (a 39 KB patch per 32-channel chunk, ~10 B/clk/CU) -- together a fifth of the
// inference step.  Here conv1_1 is evaluated tile by tile straight into conv1_2's LDS patch buffers: the 64-channel map
// never exists, conv1_2's operands need no DMA (its 72 KB of weights stay resident in LDS), and the only global traffic is
// the raw image in and the (pooled) map out.  Price: conv1_1 is recomputed on the halo (612 patch pixels for 512 outputs,
// +20 %), ~40 % more MFMAs than conv1_2 alone.
//
//   * Tile = 16x32 output pixels of conv1_2 x 64 channels, 8 waves, persistent workgroups over XCD-contiguous tile ranges,
//     150 KB of LDS: two patch buffers (conv1_1 channels 0-31 / 32-63 of the 18x34 halo patch: conv1_2's two 32-channel
//     chunks) + conv1_2's weights for both chunks (loaded once per workgroup by LDS-DMA).
//   * Phase A (conv1_1): the arithmetic of conv_first_mfma_kernel (pointwise.hip), bit for bit -- every fp32 operand split
//     exactly into three bf16 terms, K = 9 cin (+ the bias against a constant 1) in ONE v_mfma_f32_16x16x32_bf16 step, six
//     products in the same order -- on blocks of 16 consecutive patch pixels (39 blocks, wave w takes w, w + 8, ...), each
//     lane gathering its own pixel's taps (the patch rows are 34 pixels long: a block straddles rows).  Patch pixels outside
//     the image are conv1_2's zero padding: all operands zero there, the bias slot included.  The block leaves as bf16
//     through four 8-byte LDS stores per lane, in generation 4's swizzled patch layout.
//   * Phase B (conv1_2): the item loop of generation 4's 16x16 form (conv_f8_dma.hip: counted lgkmcnt fragment schedule,
//     MFMA bursts at raised priority, accumulation started from the bias, packed bf16 epilogue with the fused 2x2 max-pool)
//     over chunk 0 / buffer 0 and chunk 1 / buffer 1 -- without a single DMA or vmcnt wait.
//   * Two barriers per tile: patch written -> taps; taps done -> next patch may be written.
//   * Round 5, the RAW patch.  Switching the phases off one at a time showed phase A -- a twentieth of the FLOPs -- taking
//     135 (depth) .. 190 us (RGB) of the kernel's 330 .. 410 us, phase B 160 us, the rest 57 us: every lane gathered its 3x3
//     taps from global memory one 16-pixel block ahead (75 instructions of coordinate and mask arithmetic per block, and a
//     dependent round trip).  Now the 20 x 36-pixel raw input patch of the NEXT tile (8.6 KB of fp32 for RGB) is fetched by
//     LDS-DMA while phase B of the current tile runs, phase A gathers its taps from LDS, and tiles whose raw patch lies
//     inside the image (84 % at 768x384) take a mask-free form.  Same values, same arithmetic: bit-identical results.
//     One box, A/B (tools/first_pair_bench.py, non-integer inputs): RGB 411-418 -> 356-376 us, depth 333 -> 320 us.  Phase A
//     is still ~110 us: it is bound by its instruction count (the three-way operand split, the bf16 conversion and the
//     swizzled 8-byte LDS stores: ~120 vector instructions per 16-pixel block beside its 12-24 MFMAs, two waves per SIMD),
//     not by latency -- only running it beside phase B (a second pair of patch buffers: 78 KB the LDS does not have) would hide it.
// Results are identical to xv_conv2d_first_fwd followed by xv_conv2d_fwd (the same sums in the same order, the same bf16
// rounding of conv1_1's output): tests/test_kernels_gpu.py::test_first_pair_fused_equals_the_two_kernels.
#include <type_traits>

#include "xv_common.h"

namespace {

struct F1Args {
  const float* x;     // raw input, dense NHWC fp32 [N][H][W][CIN]
  const float* w1;    // conv1_1 kernel, fp32 HWIO [3][3][CIN][64]
  const float* b1;    // [64]
  const char* wpk2;   // conv1_2 packed weights (three images; this kernel reads the third)
  const float* b2;    // [64]
  char* y;            // bf16 [N][H+2][W+2][64] or null
  char* pooled;       // bf16 [N][H/2+2][W/2+2][64] or null
  int N, H, W;
  int tiles_x, tiles_y, n_tiles;
  int relu1, relu2;
  float out_mul;      // OF8: 2^-scale_exp of the e4m3 output maps
};

struct F1 {
  static constexpr int NWAVES = 8, NT = 512;
  static constexpr int TH = 16, TW = 32, HH = TH + 2, HW = TW + 2, NPIX = HH * HW;  // 18 x 34 = 612 patch pixels
  static constexpr int NBLK = (NPIX + 15) / 16;                                     // 39 blocks of 16 pixels
  static constexpr int A_BYTES = NBLK * 16 * 64;                                    // 39 KB (624 pixel rows of 64 B)
  static constexpr int B_PIECES = 9 * 4, B_BYTES = B_PIECES * 1024;                 // 36 KB per 32-channel chunk
  static constexpr int B_OFF = 2 * A_BYTES;
  static constexpr int LDS_BYTES = 2 * A_BYTES + 2 * B_BYTES;
  static constexpr int PROW = HW * 64;
  // the raw input patch of a tile (conv1_1's taps of the 18 x 34 halo patch): 20 x 36 pixels x CIN floats, linear, row pitch
  // 36 CIN floats; fetched by 256-byte LDS-DMA pieces (the last one padded)
  static constexpr int RAW_H = HH + 2, RAW_W = HW + 2, RAW_OFF = LDS_BYTES;
  static constexpr int raw_pieces(int cin) { return (RAW_H * RAW_W * cin + 63) / 64; }
  static constexpr int lds_bytes(int cin) { return LDS_BYTES + raw_pieces(cin) * 256; }
};
static_assert(F1::lds_bytes(3) <= 160 * 1024, "does not fit the LDS");

__device__ __forceinline__ int f1_swz(int row, int slot) { return slot ^ ((row >> 1) & 2); }  // = g4_swz16

// the k-slot map of conv_first_mfma_kernel (pointwise.hip first_k_map): k-group g < 3 = window row g in memory order (the
// first 8 of its 3 x CIN floats), group 3 = the leftovers (last channel of column 2 of the three rows) + the bias slot
template <int CIN>
__device__ __forceinline__ bool f1_k_map(int g, int e, int& dy, int& dx, int& ci) {
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

// v = h + m + l exactly, each term a bf16 (as fp32 bit patterns with zero low halves)
__device__ __forceinline__ void f1_split3(float v, uint32_t& h, uint32_t& m, uint32_t& l) {
  h = __builtin_bit_cast(uint32_t, v) & 0xffff0000u;
  const float r1 = v - __builtin_bit_cast(float, h);  // exact
  m = __builtin_bit_cast(uint32_t, r1) & 0xffff0000u;
  l = __builtin_bit_cast(uint32_t, r1 - __builtin_bit_cast(float, m));  // exact, fits 8 bits
}
__device__ __forceinline__ uint32_t f1_hi16_pair(uint32_t a, uint32_t b) {  // (a >> 16) | (b & 0xffff0000): v_perm_b32
  return __builtin_amdgcn_perm(b, a, 0x07060302u);
}
__device__ __forceinline__ void f1_split3x8(const float (&v)[8], bf16x8& h, bf16x8& m, bf16x8& l) {
  uint32_t hh[8], mm[8], ll[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) f1_split3(v[e], hh[e], mm[e], ll[e]);
  h = __builtin_bit_cast(bf16x8, u32x4{f1_hi16_pair(hh[0], hh[1]), f1_hi16_pair(hh[2], hh[3]), f1_hi16_pair(hh[4], hh[5]), f1_hi16_pair(hh[6], hh[7])});
  m = __builtin_bit_cast(bf16x8, u32x4{f1_hi16_pair(mm[0], mm[1]), f1_hi16_pair(mm[2], mm[3]), f1_hi16_pair(mm[4], mm[5]), f1_hi16_pair(mm[6], mm[7])});
  l = __builtin_bit_cast(bf16x8, u32x4{f1_hi16_pair(ll[0], ll[1]), f1_hi16_pair(ll[2], ll[3]), f1_hi16_pair(ll[4], ll[5]), f1_hi16_pair(ll[6], ll[7])});
}

// OF8: y / pooled are e4m3 maps (value * out_mul, saturating): the first e4m3 map of the fp8 graph (fcn.fp8_plan), with the
// epilogue of generation 4's <bf16 in, e4m3 out> form -- the same bytes as xv_conv2d_first_fwd + xv_conv2d_fwd onto e4m3 maps
template <int CIN, bool OF8 = false>
__global__ __launch_bounds__(512, 2) void conv_first_pair_kernel(F1Args a) {
  using C = F1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n15 = lane & 15, g = lane >> 4;
  const int H = a.H, W = a.W;
  const int Wp = W + 2;
  constexpr int Ob = OF8 ? 64 : 128;  // bytes per pixel of the output maps (64 channels of bf16 / e4m3)

  // persistent workgroups, XCD-contiguous tile ranges (as generation 2 / 4)
  const int G = gridDim.x, b = blockIdx.x;
  const int xcd = b & 7, bi = b >> 3;
  const int nb = (G - xcd + 7) >> 3;
  const int T = a.n_tiles;
  const int tq = T >> 3, trm = T & 7;
  const int t_begin = xcd * tq + (xcd < trm ? xcd : trm);
  const int t_end = t_begin + tq + (xcd < trm ? 1 : 0);
  int lid = t_begin + bi;
  if (lid >= t_end) return;

  // ---- once per workgroup: conv1_2's weights (third packed image, both 32-channel chunks) into LDS ----
  {
    const char* wimg = a.wpk2 + (int64_t)4 * 9 * 64 * 64;  // [tap][chunk 0..1][64 rows][64 B]
#pragma unroll
    for (int it = 0; it < (2 * C::B_PIECES + C::NWAVES - 1) / C::NWAVES; ++it) {
      const int piece = wave + it * C::NWAVES;  // 0 .. 71: chunk = piece / 36, then tap, then KB of the tap
      if (piece < 2 * C::B_PIECES) {
        const int chunk = piece / C::B_PIECES, q = piece - chunk * C::B_PIECES;
        const char* src = wimg + ((int64_t)chunk * 64 << 6) + (q >> 2) * (2 * 64 * 64) + (q & 3) * 1024;
        asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(C::B_OFF + chunk * C::B_BYTES + q * 1024),
                     "v"(lane * 16), "s"(src)
                     : "memory");
      }
    }
  }
  // conv1_1's weight fragments (A operand: row = channel n15 of the 16-channel block jb, k-group g), split three ways; the
  // bias rides in the first spare k slot of group 3 against a constant 1.0 on the image side
  constexpr int BIAS_E = CIN == 3 ? 3 : 0;
  bf16x8 wh[4], wm[4], wl[4];
#pragma unroll
  for (int jb = 0; jb < 4; ++jb) {
    float wv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      int dy, dx, ci;
      const bool ok = f1_k_map<CIN>(g, e, dy, dx, ci);
      wv[e] = ok ? a.w1[((dy * 3 + dx) * CIN + ci) * 64 + jb * 16 + n15] : 0.f;
      if (e == BIAS_E) wv[e] = g == 3 ? a.b1[jb * 16 + n15] : wv[e];
    }
    f1_split3x8(wv, wh[jb], wm[jb], wl[jb]);
  }
  // conv1_2's bias as four accumulator-shaped registers (lane's channels 16 g + 4 j + q: weight row 16 j + 4 g + q)
  f32x4 bvec[4];
#pragma unroll
  for (int j4 = 0; j4 < 4; ++j4) bvec[j4] = *reinterpret_cast<const f32x4*>(a.b2 + 16 * g + 4 * j4);

  // LDS fragment addresses of phase B (generation 4, 16x16 form)
  int pbase[3];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) pbase[dx] = ((2 * wave) * C::HW + n15 + dx) * 64 + (f1_swz(n15 + dx, g) << 4);
  const int wbase = C::B_OFF + n15 * 64 + (f1_swz(n15, g) << 4);

  const uint32_t floor1 = a.relu1 ? 0u : 0x80008000u;
  const bool main = g < 3;

  f32x4 acc4[2][2][2][2];  // [row i][channel pair j][pixel half h][jj]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) acc4[i][j][h][jj] = f32x4{0.f, 0.f, 0.f, 0.f};

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the weight DMA; the first barrier below publishes it)

  int n, y0, x0;
  auto decode = [&](int l, int& tn, int& ty0, int& tx0) {
    int r = l;
    tx0 = (r % a.tiles_x) * C::TW;
    r /= a.tiles_x;
    ty0 = (r % a.tiles_y) * C::TH;
    tn = r / a.tiles_y;
  };
  decode(lid, n, y0, x0);
  // ---- the raw input patch by LDS-DMA: dword d of the linear [20][36 CIN] image comes from image row y0 - 2 + d / (36 CIN);
  // offsets outside the image are clamped (what lands there is masked when it is used, as the global gather did)
  constexpr int RP = C::RAW_W * CIN, RAW_PIECES = C::raw_pieces(CIN);
  auto dma_raw = [&](int tn, int ty0, int tx0) {
    tn = __builtin_amdgcn_readfirstlane(tn), ty0 = __builtin_amdgcn_readfirstlane(ty0), tx0 = __builtin_amdgcn_readfirstlane(tx0);
    // (an "s" operand must be provably wave-uniform: the image base from two readfirstlane halves)
    const uint64_t xa = (uint64_t)(a.x + (int64_t)tn * H * W * CIN);
    const uint32_t xlo = __builtin_amdgcn_readfirstlane((uint32_t)xa), xhi = __builtin_amdgcn_readfirstlane((uint32_t)(xa >> 32));
    const float* xi = reinterpret_cast<const float*>(((uint64_t)xhi << 32) | xlo);
    const int last = H * W * CIN - 1;
#pragma unroll
    for (int it = 0; it < (RAW_PIECES + C::NWAVES - 1) / C::NWAVES; ++it) {
      const int piece = wave + it * C::NWAVES;
      if (piece < RAW_PIECES) {
        const int d = piece * 64 + lane;
        const int r = d / RP, c = d - r * RP;
        int off = ((ty0 - 2 + r) * W + tx0 - 2) * CIN + c;
        off = off < 0 ? 0 : (off > last ? last : off);
        asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dword %1, %2" ::"s"(__builtin_amdgcn_readfirstlane(C::RAW_OFF + piece * 256)),
                     "v"(off * 4), "s"(xi)
                     : "memory");
      }
    }
  };
  dma_raw(n, y0, x0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the first barrier of the loop publishes it)
  // the taps of one patch pixel per lane: block blk of the current tile, gathered from the raw patch in LDS (masked at use)
  const float* const rawp = reinterpret_cast<const float*>(smem + C::RAW_OFF);
  float raw[3][CIN];
  bool ok[3], inside;
  int pcur, hxcur;
  // FAST (a tile whose whole raw patch lies inside the image -- 84 % of the tiles of a 768x384 input): no coordinate
  // arithmetic and no masks; a lane's three taps sit at base + t * stride (stride = one pixel for the row lanes g < 3, one
  // patch row for the leftover lanes g = 3).  The general form keeps the image-coordinate masks of the global gather.
  const int lane_base = main ? g * RP : 2 * CIN, lane_stride = main ? CIN : RP;
  auto request = [&](auto fast, int blk, int tn, int ty0, int tx0) {
    const int p = blk * 16 + n15;
    const int pc = p < C::NPIX ? p : C::NPIX - 1;
    const int hy = pc / C::HW, hx = pc - hy * C::HW;
    if constexpr (decltype(fast)::value) {
      inside = p < C::NPIX;
      const float* src = rawp + hy * RP + hx * CIN + lane_base;
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int c = 0; c < CIN; ++c) raw[t][c] = src[t * lane_stride + c];
    } else {
      const int iy = ty0 - 1 + hy, ix = tx0 - 1 + hx;  // conv1_1 output pixel (image coordinates)
      inside = p < C::NPIX && iy >= 0 && iy < H && ix >= 0 && ix < W;
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const int ry = main ? iy + g - 1 : iy - 1 + t, rx = main ? ix - 1 + t : ix + 1;
        ok[t] = inside && ry >= 0 && ry < H && rx >= 0 && rx < W && (main || CIN == 3);
        // raw-patch coordinates of the same tap: row hy + (ry - iy + 1), column hx + (rx - ix + 1)
        const int prow = main ? hy + g : hy + t, pcol = main ? hx + t : hx + 2;
        const float* src = rawp + prow * RP + pcol * CIN;
#pragma unroll
        for (int c = 0; c < CIN; ++c) raw[t][c] = src[c];
      }
    }
    pcur = p;
    hxcur = hx;
  };

  while (true) {
    // every wave has finished reading the previous tile's patch buffers -- and has seen its own raw-patch DMA land (the
    // s_waitcnt vmcnt(0) in front of the previous epilogue / of the loop)
    asm volatile("s_barrier" ::: "memory");
    // the tile's raw patch lies inside the image (uniform): the mask-free form of phase A
    const bool interior = y0 >= 16 && y0 + 32 <= H && x0 >= 32 && x0 + 64 <= W;
    auto phase_a = [&](auto fast) {
      request(fast, wave, n, y0, x0);

      // ---- phase A: conv1_1 on the 18 x 34 halo patch -> bf16 -> LDS ----
      for (int blk = wave; blk < C::NBLK; blk += C::NWAVES) {
        // B operand of this block (column = pixel n15, k-group g): masked taps in k order, split three ways
        float v[8];
        if constexpr (decltype(fast)::value) {
          // (every tap is inside the image: the values of the general form with all masks true)
          const float one = inside ? 1.f : 0.f;
          if constexpr (CIN == 3) {
            v[0] = main ? raw[0][0] : raw[0][2], v[1] = main ? raw[0][1] : raw[1][2], v[2] = main ? raw[0][2] : raw[2][2];
            v[3] = main ? raw[1][0] : one;
            v[4] = main ? raw[1][1] : 0.f, v[5] = main ? raw[1][2] : 0.f, v[6] = main ? raw[2][0] : 0.f, v[7] = main ? raw[2][1] : 0.f;
          } else {
            v[0] = main ? raw[0][0] : one, v[1] = main ? raw[1][0] : 0.f, v[2] = main ? raw[2][0] : 0.f;
            v[3] = v[4] = v[5] = v[6] = v[7] = 0.f;
          }
          if (!inside) {      // (lanes past the patch in the last block: all operands zero, as in the general form)
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = 0.f;
          }
        } else if constexpr (CIN == 3) {
          // (the taps as VALUES before they meet a select: the compiler turned `main ? raw[0][0] : raw[0][2]` into one load at
          // a selected address, which put the whole array on the stack -- 32 bytes of scratch and an indexed scratch_load per
          // block in this edge-tile form, found by tools/occupancy_scan.py in round 5)
          float r[3][3];
#pragma unroll
          for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              r[t][c] = raw[t][c];
              asm volatile("" : "+v"(r[t][c]));
            }
          v[0] = ok[0] ? (main ? r[0][0] : r[0][2]) : 0.f;
          v[1] = main ? (ok[0] ? r[0][1] : 0.f) : (ok[1] ? r[1][2] : 0.f);
          v[2] = main ? (ok[0] ? r[0][2] : 0.f) : (ok[2] ? r[2][2] : 0.f);
          v[3] = main ? (ok[1] ? r[1][0] : 0.f) : (inside ? 1.f : 0.f);  // k-group 3: the bias slot (zero padding outside)
          v[4] = main && ok[1] ? r[1][1] : 0.f;
          v[5] = main && ok[1] ? r[1][2] : 0.f;
          v[6] = main && ok[2] ? r[2][0] : 0.f;
          v[7] = main && ok[2] ? r[2][1] : 0.f;
        } else {
          v[0] = main ? (ok[0] ? raw[0][0] : 0.f) : (inside ? 1.f : 0.f);  // k-group 3: the bias slot
          v[1] = ok[1] ? raw[1][0] : 0.f, v[2] = ok[2] ? raw[2][0] : 0.f;
          v[3] = v[4] = v[5] = v[6] = v[7] = 0.f;
        }
        // Operands that ARE bf16 values (raw 8-bit images as floats: the reference's rgb input, cityscapes.py:170-183) have
        // no middle / low term: the four products with them add exact zeros.  Decided per block for the whole wave.
        uint32_t lowbits = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) lowbits |= __builtin_bit_cast(uint32_t, v[e]);
        const bool exact8 = __builtin_amdgcn_ballot_w64((lowbits & 0xffffu) != 0) == 0;
        bf16x8 xh, xm, xl;
        if (exact8) {
          xh = __builtin_bit_cast(bf16x8, u32x4{f1_hi16_pair(__builtin_bit_cast(uint32_t, v[0]), __builtin_bit_cast(uint32_t, v[1])),
                                                f1_hi16_pair(__builtin_bit_cast(uint32_t, v[2]), __builtin_bit_cast(uint32_t, v[3])),
                                                f1_hi16_pair(__builtin_bit_cast(uint32_t, v[4]), __builtin_bit_cast(uint32_t, v[5])),
                                                f1_hi16_pair(__builtin_bit_cast(uint32_t, v[6]), __builtin_bit_cast(uint32_t, v[7]))});
        } else {
          f1_split3x8(v, xh, xm, xl);
        }
        const int p = pcur, hx = hxcur;
        if (blk + C::NWAVES < C::NBLK) request(fast, blk + C::NWAVES, n, y0, x0);  // next block's taps land behind this block's MFMAs
        f32x4 acc[4];
        if (exact8) {  // the three products with xh, in the order they have among the six
#pragma unroll
          for (int jb = 0; jb < 4; ++jb) acc[jb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[jb], xh, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
          for (int jb = 0; jb < 4; ++jb) acc[jb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm[jb], xh, acc[jb], 0, 0, 0);
#pragma unroll
          for (int jb = 0; jb < 4; ++jb) acc[jb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[jb], xh, acc[jb], 0, 0, 0);
        } else {
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
        }
        // lane (pixel n15, group g) holds channels 16 jb + 4 g .. + 3 of block jb: 8 bytes of chunk jb >> 1, 16-byte slot
        // 2 (jb & 1) + (g >> 1) of the pixel's 64-byte row (swizzled by the pixel's patch column), half g & 1
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) {
          typedef float f32x2 __attribute__((ext_vector_type(2)));
          typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
          const uint32_t lo = pk_max_i16(__builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{acc[jb][0], acc[jb][1]}, bf16x2)), floor1);
          const uint32_t hi = pk_max_i16(__builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{acc[jb][2], acc[jb][3]}, bf16x2)), floor1);
          const int slot = 2 * (jb & 1) + (g >> 1);
          *reinterpret_cast<u32x2*>(smem + (jb >> 1) * C::A_BYTES + p * 64 + (f1_swz(hx, slot) << 4) + 8 * (g & 1)) = u32x2{lo, hi};
        }
      }
    };
    if (interior)
      phase_a(std::true_type{});
    else
      phase_a(std::false_type{});
    // the patch is complete: this wave's LDS stores have landed, then everybody's
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // (nobody reads the raw patch any more:) the next tile's raw patch travels during phase B
    const int nlid = lid + nb;
    const bool has_next = nlid < t_end;
    int n2 = n, y02 = y0, x02 = x0;
    if (has_next) {
      decode(nlid, n2, y02, x02);
      dma_raw(n2, y02, x02);
    }

    // ---- phase B: conv1_2, two items (chunk c = patch buffer c = weight buffer c), generation 4's 16x16 schedule ----
#pragma unroll
    for (int chunk = 0; chunk < 2; ++chunk) {
      const int pb0[3] = {pbase[0] + chunk * C::A_BYTES, pbase[1] + chunk * C::A_BYTES, pbase[2] + chunk * C::A_BYTES};
      const int wb0 = wbase + chunk * C::B_BYTES;
      u32x4 wf[2][2][2], xf[2][4][2];
#define F1_RD(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
      // t = dx*3 + dy  ->  packed tap dy*3 + dx; fragment (j, jj) = channel block 2 j + jj, 1 KB apart
#define F1_LDW(t, set)                                      \
  {                                                         \
    constexpr int tap_ = (((t) % 3) * 3 + (t) / 3) * 4096;  \
    F1_RD(wf[set][0][0], wb0, tap_);                        \
    F1_RD(wf[set][0][1], wb0, tap_ + 1024);                 \
    F1_RD(wf[set][1][0], wb0, tap_ + 2048);                 \
    F1_RD(wf[set][1][1], wb0, tap_ + 3072);                 \
  }
#define F1_LDPA(dx, set)                            \
  {                                                 \
    F1_RD(xf[set][0][0], pb0[dx], 0);               \
    F1_RD(xf[set][0][1], pb0[dx], 1024);            \
    F1_RD(xf[set][1][0], pb0[dx], C::PROW);         \
    F1_RD(xf[set][1][1], pb0[dx], C::PROW + 1024);  \
  }
#define F1_LDPB(dx, set)                                \
  {                                                     \
    F1_RD(xf[set][2][0], pb0[dx], 2 * C::PROW);         \
    F1_RD(xf[set][2][1], pb0[dx], 2 * C::PROW + 1024);  \
    F1_RD(xf[set][3][0], pb0[dx], 3 * C::PROW);         \
    F1_RD(xf[set][3][1], pb0[dx], 3 * C::PROW + 1024);  \
  }
#define F1_WAIT_W(n, ws, ps) \
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(wf[ws][0][0]), "+v"(wf[ws][0][1]), "+v"(wf[ws][1][0]), "+v"(wf[ws][1][1]) : "n"(n))
#define F1_WAIT_WA(n, ws, ps)                                                                                         \
  asm volatile("s_waitcnt lgkmcnt(%8)"                                                                                \
               : "+v"(wf[ws][0][0]), "+v"(wf[ws][0][1]), "+v"(wf[ws][1][0]), "+v"(wf[ws][1][1]), "+v"(xf[ps][0][0]),  \
                 "+v"(xf[ps][0][1]), "+v"(xf[ps][1][0]), "+v"(xf[ps][1][1])                                           \
               : "n"(n))
#define F1_WAIT_WB(n, ws, ps)                                                                                         \
  asm volatile("s_waitcnt lgkmcnt(%8)"                                                                                \
               : "+v"(wf[ws][0][0]), "+v"(wf[ws][0][1]), "+v"(wf[ws][1][0]), "+v"(wf[ws][1][1]), "+v"(xf[ps][2][0]),  \
                 "+v"(xf[ps][2][1]), "+v"(xf[ps][3][0]), "+v"(xf[ps][3][1])                                           \
               : "n"(n))
#define F1_C_ACC(i, j, h, jj) acc4[i][j][h][jj]
#define F1_C_BIAS(i, j, h, jj) bvec[2 * (j) + (jj)]
#define F1_MFMA_C(i, j, ws, ps, dy, CS)                                                                                \
  {                                                                                                                    \
    acc4[i][j][0][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[ws][j][0]),               \
                                                               __builtin_bit_cast(bf16x8, xf[ps][(i) + (dy)][0]), CS(i, j, 0, 0), 0, 0, 0); \
    acc4[i][j][0][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[ws][j][1]),               \
                                                               __builtin_bit_cast(bf16x8, xf[ps][(i) + (dy)][0]), CS(i, j, 0, 1), 0, 0, 0); \
    acc4[i][j][1][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[ws][j][0]),               \
                                                               __builtin_bit_cast(bf16x8, xf[ps][(i) + (dy)][1]), CS(i, j, 1, 0), 0, 0, 0); \
    acc4[i][j][1][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[ws][j][1]),               \
                                                               __builtin_bit_cast(bf16x8, xf[ps][(i) + (dy)][1]), CS(i, j, 1, 1), 0, 0, 0); \
  }
#define F1_PIN(i, j) \
  asm volatile("" : "+v"(acc4[i][j][0][0]), "+v"(acc4[i][j][0][1]), "+v"(acc4[i][j][1][0]), "+v"(acc4[i][j][1][1]))
#define F1_TAP_BODY(t, CS)                                        \
  {                                                               \
    constexpr int dx_ = (t) / 3, dy_ = (t) % 3;                   \
    F1_MFMA_C(0, 0, (t) & 1, dx_ & 1, dy_, CS);                   \
    F1_PIN(0, 0);                                                 \
    __builtin_amdgcn_s_setprio(2);                                \
    __builtin_amdgcn_sched_barrier(0);                            \
    F1_MFMA_C(0, 1, (t) & 1, dx_ & 1, dy_, CS);                   \
    F1_MFMA_C(1, 0, (t) & 1, dx_ & 1, dy_, CS);                   \
    F1_MFMA_C(1, 1, (t) & 1, dx_ & 1, dy_, CS);                   \
    F1_PIN(0, 1);                                                 \
    F1_PIN(1, 0);                                                 \
    F1_PIN(1, 1);                                                 \
    __builtin_amdgcn_sched_barrier(0);                            \
    __builtin_amdgcn_s_setprio(1);                                \
    __builtin_amdgcn_sched_barrier(0);                            \
  }
#define F1_TAP(t, WAIT, NEWER, POST)                              \
  {                                                               \
    if constexpr ((t) + 1 < 9) F1_LDW((t) + 1, ((t) + 1) & 1);    \
    WAIT(NEWER, (t) & 1, ((t) / 3) & 1);                          \
    POST;                                                         \
    __builtin_amdgcn_sched_barrier(0);                            \
    F1_TAP_BODY(t, F1_C_ACC)                                      \
  }
      F1_LDW(0, 0);
      F1_LDPA(0, 0);
      __builtin_amdgcn_sched_barrier(0);
      // tap 0: W0 + Pa0 in flight; W1 behind the wait; the tile's first MFMAs start from the bias
      F1_WAIT_WA(0, 0, 0);
      F1_LDW(1, 1);
      F1_LDPB(0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (chunk == 0) {
        F1_TAP_BODY(0, F1_C_BIAS)
      } else {
        F1_TAP_BODY(0, F1_C_ACC)
      }
      F1_TAP(1, F1_WAIT_WB, 4, F1_LDPA(1, 1))
      F1_TAP(2, F1_WAIT_W, 8, F1_LDPB(1, 1))
      F1_TAP(3, F1_WAIT_WA, 8, )
      F1_TAP(4, F1_WAIT_WB, 4, F1_LDPA(2, 0))
      F1_TAP(5, F1_WAIT_W, 8, F1_LDPB(2, 0))
      F1_TAP(6, F1_WAIT_WA, 8, )
      F1_TAP(7, F1_WAIT_WB, 4, )
      F1_TAP(8, F1_WAIT_W, 0, )
#undef F1_RD
#undef F1_LDW
#undef F1_LDPA
#undef F1_LDPB
#undef F1_WAIT_W
#undef F1_WAIT_WA
#undef F1_WAIT_WB
#undef F1_C_ACC
#undef F1_C_BIAS
#undef F1_MFMA_C
#undef F1_PIN
#undef F1_TAP_BODY
#undef F1_TAP
    }

    // this wave's share of the next raw patch has landed (requested a whole phase B ago; the tile's stores come after it)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- tile epilogue (generation 4's packed form): bf16 pairs, relu and the 2x2 max on signed 16-bit integers ----
    const int py = y0 + 2 * wave;
    const int cofs = 16 * g;  // first of this lane's 16 consecutive channels
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int px = x0 + 16 * u + n15;
      if constexpr (OF8) {
        // e4m3 epilogue (conv_f8_dma.hip, the 16x16 form: the bias is inside the accumulators): scale by a power of two,
        // relu + saturation in one v_med3_f32, the 2x2 max as v_med3_f32(x, y, +inf), one conversion
        const float lo = a.relu2 ? 0.f : -448.f, inf = __builtin_inff();
        float q[2][16];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            q[i][r] = __builtin_amdgcn_fmed3f(acc4[i][r >> 3][u][(r >> 2) & 1][r & 3] * a.out_mul, lo, 448.f);
        auto cvt16 = [&](const float (&w)[16]) {
          u32x4 o;
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            int p = 0;
            p = __builtin_amdgcn_cvt_pk_fp8_f32(w[4 * d], w[4 * d + 1], p, false);
            p = __builtin_amdgcn_cvt_pk_fp8_f32(w[4 * d + 2], w[4 * d + 3], p, true);
            o[d] = (uint32_t)p;
          }
          return o;
        };
        if (a.y != nullptr) {
#pragma unroll
          for (int i = 0; i < 2; ++i)
            *reinterpret_cast<u32x4*>(a.y + (((int64_t)n * (H + 2) + (py + i + 1)) * Wp + (px + 1)) * Ob + cofs) = cvt16(q[i]);
        }
        if (a.pooled != nullptr) {
          float m[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float t = __builtin_amdgcn_fmed3f(q[0][r], q[1][r], inf);
            const float o = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, t), 0xB1, 0xf, 0xf, true));
            m[r] = __builtin_amdgcn_fmed3f(t, o, inf);
          }
          const int Hq = H >> 1, Wq = W >> 1;
          const u32x4 o = cvt16(m);
          if ((lane & 1) == 0)
            *reinterpret_cast<u32x4*>(a.pooled + (((int64_t)n * (Hq + 2) + ((py >> 1) + 1)) * (Wq + 2) + ((px >> 1) + 1)) * Ob + cofs) = o;
        }
        continue;
      }
      uint32_t pk[2][8];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int k = 0; k < 8; ++k) {  // channels 2 k, 2 k + 1 of the lane's 16: r = 4 (2 j + jj) + q
          const int r0 = 2 * k, r1 = 2 * k + 1;
          pk[i][k] = pack_bf16x2(acc4[i][r0 >> 3][u][(r0 >> 2) & 1][r0 & 3], acc4[i][r1 >> 3][u][(r1 >> 2) & 1][r1 & 3]);
        }
      if (a.y != nullptr) {
        const uint32_t rfloor = a.relu2 ? 0u : 0x80008000u;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
          for (int k = 0; k < 8; ++k) pk[i][k] = pk_max_i16(pk[i][k], rfloor);
          char* dst = a.y + (((int64_t)n * (H + 2) + (py + i + 1)) * Wp + (px + 1)) * Ob + cofs * 2;
          *reinterpret_cast<u32x4*>(dst) = u32x4{pk[i][0], pk[i][1], pk[i][2], pk[i][3]};
          *reinterpret_cast<u32x4*>(dst + 16) = u32x4{pk[i][4], pk[i][5], pk[i][6], pk[i][7]};
        }
      }
      if (a.pooled != nullptr) {
        uint32_t m[8];
        if (a.relu2) {
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const uint32_t t = pk_max_i16(pk[0][k], pk[1][k]);
            m[k] = pk_max_i16(pk_max_i16(t, pk_dpp_swap1(t)), 0u);
          }
        } else {
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const uint32_t t = pk_max_i16(pk_ord_bf16(pk[0][k]), pk_ord_bf16(pk[1][k]));
            m[k] = pk_ord_bf16(pk_max_i16(t, pk_dpp_swap1(t)));
          }
        }
        const int Hq = H >> 1, Wq = W >> 1;
        char* dst = a.pooled + (((int64_t)n * (Hq + 2) + ((py >> 1) + 1)) * (Wq + 2) + ((px >> 1) + 1)) * Ob + cofs * 2;
        if ((lane & 1) == 0) {
          *reinterpret_cast<u32x4*>(dst) = u32x4{m[0], m[1], m[2], m[3]};
          *reinterpret_cast<u32x4*>(dst + 16) = u32x4{m[4], m[5], m[6], m[7]};
        }
      }
    }
    if (!has_next) break;
    lid = nlid;
    n = n2, y0 = y02, x0 = x02;
  }
}

template <int CIN, bool OF8 = false>
int f1_launch(const F1Args& a, int grid, hipStream_t stream) {
  static bool attr_set[XV_MAX_DEVICES] = {false};
  const hipError_t e =
      xv_allow_dynamic_lds(reinterpret_cast<const void*>(&conv_first_pair_kernel<CIN, OF8>), F1::lds_bytes(CIN), attr_set);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL((conv_first_pair_kernel<CIN, OF8>), dim3((unsigned)grid), dim3(F1::NT), F1::lds_bytes(CIN), stream, a);
  return xv_launch_status();
}

}  // namespace

// conv1_1 (raw fp32 input, fp32 HWIO weights) followed by conv1_2 (packed bf16 weights, 64 -> 64 channels) in one launch;
// y and / or pooled as xv_conv2d_fwd.  Maps that tile exactly in 16x32, 1 or 3 input channels; XV_ESHAPE otherwise (the
// caller then runs the two kernels).
extern "C" int xv_conv_first_pair_fwd(const float* x, int n, int h, int w, int cin, const float* w1_hwio, const float* b1, int relu1,
                                      const void* w2_packed, const float* b2, int relu2, const xv_act* y, const xv_act* pooled,
                                      void* stream) {
  XV_CHECK_ARG(x && w1_hwio && b1 && w2_packed && b2);
  const bool has_y = y && y->data, has_q = pooled && pooled->data;
  XV_CHECK_ARG(has_y || has_q);
  // bf16 maps, or e4m3 maps (the first e4m3 map of the fp8 graph): both outputs of one dtype and scale
  const xv_act* const o = has_q ? pooled : y;
  XV_CHECK_ARG(o->dtype == XV_BF16 || o->dtype == XV_FP8);
  if (has_y && has_q) XV_CHECK_ARG(y->dtype == pooled->dtype && y->scale_exp == pooled->scale_exp);
  const bool of8 = o->dtype == XV_FP8;
  if (of8) XV_CHECK_SHAPE(o->scale_exp > -100 && o->scale_exp < 100);
  XV_CHECK_SHAPE(xv_dims_sane(n, h, w) && (cin == 1 || cin == 3) && (h & 15) == 0 && (w & 31) == 0);
  XV_CHECK_SHAPE((int64_t)n * h * w * cin < 0x7ff00000);
  if (has_y) XV_CHECK_SHAPE(y->n == n && y->h == h && y->w == w && y->c == 64);
  if (has_q) XV_CHECK_SHAPE(pooled->n == n && pooled->h == h / 2 && pooled->w == w / 2 && pooled->c == 64);
  XV_CHECK_ARG((((uintptr_t)x | (uintptr_t)w1_hwio | (uintptr_t)b1 | (uintptr_t)w2_packed | (uintptr_t)b2) & 15) == 0);
  if (has_y) XV_CHECK_ARG(((uintptr_t)y->data & 15) == 0);
  if (has_q) XV_CHECK_ARG(((uintptr_t)pooled->data & 15) == 0);
  F1Args a{};
  a.x = x, a.w1 = w1_hwio, a.b1 = b1, a.wpk2 = (const char*)w2_packed, a.b2 = b2;
  a.y = has_y ? (char*)y->data : nullptr;
  a.pooled = has_q ? (char*)pooled->data : nullptr;
  a.N = n, a.H = h, a.W = w;
  a.tiles_x = w / F1::TW, a.tiles_y = h / F1::TH;
  const int64_t ntiles = (int64_t)a.tiles_x * a.tiles_y * n;
  XV_CHECK_SHAPE(ntiles > 0 && ntiles <= 0x7fffffff);
  a.n_tiles = (int)ntiles;
  a.relu1 = relu1, a.relu2 = relu2;
  a.out_mul = of8 ? exp2f((float)-o->scale_exp) : 1.f;
  const int grid = xv_num_cus();
  if (of8) return cin == 3 ? f1_launch<3, true>(a, grid, (hipStream_t)stream) : f1_launch<1, true>(a, grid, (hipStream_t)stream);
  return cin == 3 ? f1_launch<3>(a, grid, (hipStream_t)stream) : f1_launch<1>(a, grid, (hipStream_t)stream);
}
