// Generation 5: the generation-4 item loop (conv_f8_dma.hip: every operand by LDS-DMA, one barrier per item, counted
// vmcnt / lgkmcnt, v_mfma_f32_16x16x32_bf16) on a COLUMN of waves -- configuration 27 (24x16 pixel tile, 3 image rows per
// wave) and 28 (32x16, 4 rows per wave).
//
// Replaces, like the other conv kernels, tf.layers.conv2d(3x3, 'same', relu) (+ max_pooling2d) of the FCN trunk
// (xview/models/simple_fcn.py:39-79), here first of all conv5_1 .. conv5_3 of a 768x384 input: 24x48 maps.  The 16x32 tile of
// generation 4 covers such a map at 56 % (2 x 2 tiles over 32 x 64 pixels), so round 3 ran them on generation 2's 24x16
// tile (configuration 22: 0.38 of the MFMA peak, the kernel furthest below its roof).  24x16 tiles the
// map exactly; this kernel gives that tile the lean loop.
//
//   * 8 waves, wave w owns image rows MT w .. MT w + MT - 1 x 16 columns x 64 output channels = MT pixel blocks x 4 channel
//     blocks of 16x16 (16 MT accumulator registers).  Lane l = (column n15 = l & 15, k-group / channel group g = l >> 4).
//   * A tap is 4 MT MFMAs of K = 32 (the whole 32-channel chunk).  Per tap the wave reads 4 weight fragments, per column
//     group (3 taps) MT + 2 pixel fragments: 36 + 3 (MT + 2) ds_read_b128 per item against 36 + 24 for the 2 rows x 32
//     columns of generation 4 -- 54 instead of 60 at MT = 4 for the same 144 MFMAs.
//   * LDS images, the source-side swizzle of the DMA (g4_swz16: slot s of row r at s ^ ((r >> 1) & 2), conflict-free for
//     every column offset), the packed weight image (the THIRD image of xv_pack_conv_weights: rows permuted so that a
//     lane's 16 accumulators of a pixel are 16 consecutive channels) and the epilogue are generation 4's 16x16 form.
//   * Fragment schedule (reads return in order; at most 4 + (MT + 2) + 4 = 14 in flight, the counter holds 15):
//       before tap 0: bias W0 P0 | tap 0: wait, W1 | tap t: W(t+1), wait, [P of the next column group at taps 1 and 3], MFMAs.
//   * Exact tilings only (H % (8 MT) == 0, W % 16 == 0); fused 2x2 max-pool for even MT; bias + relu, no addend / mask.
//   * Round 5, CHUNK GROUPS (MODE 1 / 2): on the shapes only this kernel serves (maps that tile in 24x16 and not in 16x32: the
//     24x48 conv5 maps) with eight or more chunks, a tile's sum over input channels is taken in groups of two chunks --
//     group 0 starts from the bias, the others from zero, and the groups are added in order, out = ((g0 + g1) + g2) + ...
//     MODE 1 does that in registers.  MODE 2 (SPLIT: fewer tiles than half the CUs -- conv5_x at one image is 24 tiles
//     for 256 CUs, 38 us a layer at 142 TFLOP/s) makes every (tile, group) a work item of its own that leaves its fp32
//     accumulators in a workspace slab, and splitk_reduce_kernel adds the slabs in the same order: the same bits as MODE 1
//     whatever the batch size, so an image alone and in a batch still agree bit for bit.
#include "xv_common.h"

namespace {

struct G5Args {
  const char* x;      // bf16 [N][H+2][W+2][Cin], zero border
  const char* wpk;    // packed bf16 weights (three images; this kernel reads the third)
  const float* bias;  // [Cout]
  char* y;            // bf16 [N][H+2][W+2][Cout] or null
  char* pooled;       // bf16 [N][H/2+2][W/2+2][Cout] or null
  int N, H, W, Cin, Cout;
  int tiles_x, tiles_y, n_ct, n_tiles;
  int relu;
  // data-gradient epilogue: y = (conv + addend) where mask > 0, else 0 -- both maps shaped like y (either may be null)
  const char* mask;
  const char* addend;
  // second problem of the same shape in the same launch (conv_f8_dma.hip F8Args::n_first): the conv5 maps of a 768x384 input
  // at 16 images make 2 x 384 tiles = 3 rounds of 256 workgroups instead of 2 x 1.5 -> 2 x 2.  Forward form only.
  const char* x2;
  const char* wpk2;
  const float* bias2;
  char* y2;
  char* pooled2;
  int n_first;
  // chunk groups (MODE 1 / 2): grp chunks per group, ngrp groups per tile; MODE 2: n_tiles counts (tile, group) items and
  // slabs holds their fp32 accumulators, [item][pixel of the tile][64 channels]
  int grp, ngrp;
  float* slabs;
};

template <int MT>
struct G5 {
  static constexpr int NWAVES = 8, NT = 512;
  static constexpr int TH = NWAVES * MT, TW = 16, HH = TH + 2, HW = TW + 2, NPIX = HH * HW;
  static constexpr int R = MT + 2;                       // patch rows a wave reads per column group
  static constexpr int A_PIECES = (NPIX * 4 + 63) / 64;  // 1 KB per DMA wave-instruction
  static constexpr int A_BYTES = A_PIECES * 1024;
  static constexpr int B_PIECES = 9 * 4;  // 9 taps x (64 rows x 64 B)
  static constexpr int B_BYTES = B_PIECES * 1024;
  static constexpr int BIAS_OFF = 2 * (A_BYTES + B_BYTES);  // two 256-byte bias slots (tile parity)
  static constexpr int LDS_BYTES = BIAS_OFF + 512;
  static constexpr int A_ITERS = (A_PIECES + NWAVES - 1) / NWAVES;
  static constexpr int B_ITERS = (B_PIECES + NWAVES - 1) / NWAVES;
  static constexpr int PROW = HW * 64;  // bytes between patch rows
  static_assert(LDS_BYTES <= 160 * 1024, "does not fit the LDS");
  static_assert(R == 5 || R == 6, "the fragment macros are written for 5 or 6 patch rows");
};

__device__ __forceinline__ int g5_swz(int row, int slot) { return slot ^ ((row >> 1) & 2); }  // = g4_swz16

#ifdef XV_CLOCK_STAMP
__device__ unsigned long long xv_clk_g5[4 * XV_CLK_SLOTS];
#endif

// DG: the data-gradient epilogue (addend + relu mask) as a kernel of its own (the forward kernel carries neither its registers
// nor its branches)
template <int MT, bool DG = false, int MODE = 0>
__global__ __launch_bounds__(512, 2) void conv_dma5_kernel(G5Args a) {
  static_assert(MODE == 0 || !DG, "chunk groups exist in the forward kernel only");
  using C = G5<MT>;
  constexpr int R = C::R;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n15 = lane & 15, g = lane >> 4;
  const int H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout;
  const int Wp = W + 2;
  const int Cb = 2 * Cin, Ob = 2 * Cout;  // bytes per pixel of the input / output maps
  const int nchunks = Cb >> 6;            // 64 bytes = 32 input channels per chunk
  const char* const wimg = a.wpk + (int64_t)4 * 9 * Cin * Cout;  // the third packed image
  const char* const wimg2 = a.wpk2 + (int64_t)4 * 9 * Cin * Cout;

  // persistent workgroups, XCD-contiguous tile ranges, the output-channel tiles of one patch adjacent (as generation 2 / 4)
  const int G = gridDim.x, b = blockIdx.x;
  const int xcd = b & 7, bi = b >> 3;
  const int nb = (G - xcd + 7) >> 3;
  const int T = a.n_tiles;
  const int tq = T >> 3, trm = T & 7;
  const int t_begin = xcd * tq + (xcd < trm ? xcd : trm);
  const int t_end = t_begin + tq + (xcd < trm ? 1 : 0);
  int lid = t_begin + bi;
  if (lid >= t_end) return;

  struct Tile {
    int n, y0, x0, co0;
    int second;  // 1: a tile of the second problem (n counts its images).  (int: a bool member sent the struct through an
                 // alloca that the compiler promoted into LDS, on top of the patch buffer)
    int split;   // MODE 2: the chunk group this item computes
  };
  auto decode = [&](int l) {
    Tile t;
    t.split = 0;
    if constexpr (MODE == 2) {
      t.split = l % a.ngrp;
      l /= a.ngrp;
    }
    t.co0 = (l % a.n_ct) * 64;
    int r = l / a.n_ct;
    t.x0 = (r % a.tiles_x) * C::TW;
    r /= a.tiles_x;
    t.y0 = (r % a.tiles_y) * C::TH;
    t.n = r / a.tiles_y;
    t.second = (a.n_first > 0 && t.n >= a.n_first) ? 1 : 0;
    t.n -= t.second ? a.n_first : 0;
    return t;
  };

  // per-lane source offsets (bytes, relative to the patch origin) of the patch pieces this wave moves: LDS granule
  // gr = piece * 64 + lane holds physical slot gr & 3 of pixel gr >> 2
  int aoff[C::A_ITERS];
#pragma unroll
  for (int it = 0; it < C::A_ITERS; ++it) {
    const int gr = (wave + it * C::NWAVES) * 64 + lane;
    int p = gr >> 2;
    p = p < C::NPIX ? p : C::NPIX - 1;
    const int hy = p / C::HW, hx = p - hy * C::HW;
    aoff[it] = (hy * Wp + hx) * Cb + (g5_swz(hx, gr & 3) << 4);
  }
  // LDS fragment addresses: pixel column n15 + dx of patch row MT * wave, weight row n15 of a 16-row block; k-group g
  int pbase[3];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) pbase[dx] = ((MT * wave) * C::HW + n15 + dx) * 64 + (g5_swz(n15 + dx, g) << 4);
  const int wbase = 2 * C::A_BYTES + n15 * 64 + (g5_swz(n15, g) << 4);
  const int64_t tap_pitch = (int64_t)nchunks * Cout * 64;  // bytes between taps of the packed image

  // LDS-DMA in assembly (SGPR base + 32-bit VGPR offset; M0 = the wave's LDS destination), as generation 2 / 4
  auto dma16 = [&](const char* sbase, int voff, int lds_off) {
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_off), "v"(voff), "s"(sbase) : "memory");
  };
  auto dma_bases = [&](const Tile& t, int chunk, const char*& xsrc, const char*& wsrc) {
    xsrc = (t.second ? a.x2 : a.x) + (((int64_t)t.n * (H + 2) + t.y0) * Wp + t.x0) * Cb + chunk * 64;
    wsrc = (t.second ? wimg2 : wimg) + (((int64_t)chunk * Cout + t.co0) << 6);
  };
  auto dma_a = [&](const char* xsrc, int it, int buf) {
    const int piece = wave + it * C::NWAVES;
    if (piece < C::A_PIECES) dma16(xsrc, aoff[it], buf * C::A_BYTES + piece * 1024);
  };
  auto dma_b = [&](const char* wsrc, int it, int buf) {
    const int piece = wave + it * C::NWAVES;
    if (piece < C::B_PIECES)
      dma16(wsrc + (piece >> 2) * tap_pitch + (piece & 3) * 1024, lane * 16, 2 * C::A_BYTES + buf * C::B_BYTES + piece * 1024);
  };
  // the tile's 64 bias values ride along with its first chunk (one 4-byte-per-lane DMA by the last wave)
  auto dma_bias = [&](const Tile& t, bool tile_start, int bslot) {
    if (tile_start && wave == C::NWAVES - 1)
      asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dword %1, %2" ::"s"(C::BIAS_OFF + bslot * 256), "v"(lane * 4),
                   "s"((t.second ? a.bias2 : a.bias) + t.co0)
                   : "memory");
  };

  Tile cur = decode(lid);
  int chunk = MODE == 2 ? cur.split * a.grp : 0, buf = 0, bslot = 0;
  {
    const char *xsrc, *wsrc;
    dma_bases(cur, chunk, xsrc, wsrc);
#pragma unroll
    for (int it = 0; it < C::A_ITERS; ++it) dma_a(xsrc, it, 0);
#pragma unroll
    for (int it = 0; it < C::B_ITERS; ++it) dma_b(wsrc, it, 0);
    dma_bias(cur, true, 0);
  }

  f32x4 acc[MT][4];
  f32x4 tot[MODE == 1 ? MT : 1][4];      // MODE 1: the sum of the finished chunk groups
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < (MODE == 1 ? MT : 1); ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) tot[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // RESIDENT WEIGHTS (as generation 2 / 4): with one or two chunks per tile weight buffer p only ever holds chunk p's
  // weights -- of the same output-channel tile too when every workgroup of the XCD group keeps its tile residue
  const bool resident = MODE != 2 && nchunks <= 2 && (nb % a.n_ct) == 0 && a.n_first == 0;  // (two problems: two sets of weights)
  int items_done = 0;
  // 16-byte store instructions per tile and wave: two per row, two per pooled row
  const int nstores = MODE == 2 ? 4 * MT : (a.y != nullptr ? 2 * MT : 0) + (a.pooled != nullptr ? MT : 0);
  int in_flight = 0;  // stores issued after the last DMA of the previous item

  XV_CLK_BEGIN()
  while (true) {
    // This item's operands have landed (each wave retires its own DMA; the tile stores issued after it may stay in
    // flight: vmcnt counts in issue order), and every wave has finished reading the other buffer pair.
    if (in_flight == 4 * MT)
      asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(4 * MT) : "memory");
    else if (in_flight == 3 * MT)
      asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(3 * MT) : "memory");
    else if (in_flight == 2 * MT)
      asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * MT) : "memory");
    else if (in_flight == MT)
      asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(MT) : "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    in_flight = 0;

    const int pb[3] = {pbase[0] + buf * C::A_BYTES, pbase[1] + buf * C::A_BYTES, pbase[2] + buf * C::A_BYTES};
    const int wb = wbase + buf * C::B_BYTES;

    // Fragment registers: weights of tap t in set t & 1 ([channel block]), the R patch rows of column group dx in set dx & 1
    u32x4 wf[2][4], xf[2][6];
#define G5_RD(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
    // t = dx*3 + dy  ->  packed tap dy*3 + dx
#define G5_LDW(t, set)                                      \
  {                                                         \
    constexpr int tap_ = (((t) % 3) * 3 + (t) / 3) * 4096;  \
    G5_RD(wf[set][0], wb, tap_);                            \
    G5_RD(wf[set][1], wb, tap_ + 1024);                     \
    G5_RD(wf[set][2], wb, tap_ + 2048);                     \
    G5_RD(wf[set][3], wb, tap_ + 3072);                     \
  }
#define G5_LDP(dx, set)                                              \
  {                                                                  \
    G5_RD(xf[set][0], pb[dx], 0);                                    \
    G5_RD(xf[set][1], pb[dx], C::PROW);                              \
    G5_RD(xf[set][2], pb[dx], 2 * C::PROW);                          \
    G5_RD(xf[set][3], pb[dx], 3 * C::PROW);                          \
    if constexpr (R > 4) G5_RD(xf[set][4], pb[dx], 4 * C::PROW);     \
    if constexpr (R > 5) G5_RD(xf[set][5], pb[dx], 5 * C::PROW);     \
  }
    // at most n newer reads outstanding.  Every wait names exactly the registers it releases: the MFMAs that consume them
    // cannot move above it, and no register with a read still in flight is an operand of anything.
#define G5_WAIT_W(n, ws, ps) \
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(wf[ws][0]), "+v"(wf[ws][1]), "+v"(wf[ws][2]), "+v"(wf[ws][3]) : "n"(n))
#define G5_WAIT_WP(n, ws, ps)                                                                                          \
  if constexpr (R == 5)                                                                                                \
    asm volatile("s_waitcnt lgkmcnt(%9)"                                                                               \
                 : "+v"(wf[ws][0]), "+v"(wf[ws][1]), "+v"(wf[ws][2]), "+v"(wf[ws][3]), "+v"(xf[ps][0]), "+v"(xf[ps][1]), \
                   "+v"(xf[ps][2]), "+v"(xf[ps][3]), "+v"(xf[ps][4])                                                   \
                 : "n"(n));                                                                                            \
  else                                                                                                                 \
    asm volatile("s_waitcnt lgkmcnt(%10)"                                                                              \
                 : "+v"(wf[ws][0]), "+v"(wf[ws][1]), "+v"(wf[ws][2]), "+v"(wf[ws][3]), "+v"(xf[ps][0]), "+v"(xf[ps][1]), \
                   "+v"(xf[ps][2]), "+v"(xf[ps][3]), "+v"(xf[ps][4]), "+v"(xf[ps][5])                                  \
                 : "n"(n))
#define G5_C_ACC(i, j) acc[i][j]
    // first tap of a tile: the accumulation STARTS from the bias (lane's channels 16 g + 4 j + q) -- no bias add and no
    // accumulator clearing in the epilogue (conv_f8_dma.hip, 16x16 form)
#define G5_C_BIAS(i, j) __builtin_bit_cast(f32x4, bvec[j])
#define G5_MFMA(i, ws, ps, dy) G5_MFMA_C(i, ws, ps, dy, G5_C_ACC)
#define G5_MFMA_B(i, ws, ps, dy) G5_MFMA_C(i, ws, ps, dy, G5_C_BIAS)
#define G5_MFMA_C(i, ws, ps, dy, CS)                                                                                 \
  {                                                                                                                  \
    acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[ws][0]),                       \
                                                        __builtin_bit_cast(bf16x8, xf[ps][(i) + (dy)]), CS(i, 0), 0, 0, 0); \
    acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[ws][1]),                       \
                                                        __builtin_bit_cast(bf16x8, xf[ps][(i) + (dy)]), CS(i, 1), 0, 0, 0); \
    acc[i][2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[ws][2]),                       \
                                                        __builtin_bit_cast(bf16x8, xf[ps][(i) + (dy)]), CS(i, 2), 0, 0, 0); \
    acc[i][3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[ws][3]),                       \
                                                        __builtin_bit_cast(bf16x8, xf[ps][(i) + (dy)]), CS(i, 3), 0, 0, 0); \
  }
    // An MFMA is a pure value to the optimizer: an empty volatile asm that "modifies" its accumulators pins the MFMAs that
    // produced them in front of every later asm statement (reads, waits, priority changes), as in generation 4.
#define G5_PIN(i) asm volatile("" : "+v"(acc[i][0]), "+v"(acc[i][1]), "+v"(acc[i][2]), "+v"(acc[i][3]))
    // the next item's DMA: patch pieces two per tap from tap 0 (HBM / Infinity-Cache latency), weight pieces (L2) after,
    // the bias with the first weight piece
#define G5_DMA_PIECES(t)                                                                              \
  if (has_next) {                                                                                     \
    if (2 * (t) < C::A_ITERS) dma_a(nx_src, 2 * (t), buf ^ 1);                                        \
    if (2 * (t) + 1 < C::A_ITERS) dma_a(nx_src, 2 * (t) + 1, buf ^ 1);                                \
    if ((t) >= A_TAPS && (t) - A_TAPS < C::B_ITERS && !skip_b) dma_b(nw_src, (t) - A_TAPS, buf ^ 1);  \
    if ((t) == A_TAPS) dma_bias(nxt, last_chunk, bslot ^ 1);                                          \
  }
#define G5_TAP(t, WAIT, NEWER, POST)   \
  {                                    \
    G5_TAP_HEAD(t, WAIT, NEWER, POST)  \
    G5_TAP_BODY(t, G5_MFMA)            \
  }
#define G5_TAP_HEAD(t, WAIT, NEWER, POST)                         \
  {                                                               \
    if constexpr ((t) + 1 < 9) G5_LDW((t) + 1, ((t) + 1) & 1);    \
    G5_DMA_PIECES(t)                                              \
    WAIT(NEWER, (t) & 1, ((t) / 3) & 1);                          \
    POST;                                                         \
    __builtin_amdgcn_sched_barrier(0);                            \
  }
    // tap 0: bias + W0 + P0 = 4 + 4 + (MT + 2) <= 14 reads in flight; W1 only behind the wait for them
#define G5_TAP_HEAD0(WAIT)                                        \
  {                                                               \
    G5_DMA_PIECES(0)                                              \
    WAIT(0, 0, 0);                                                \
    G5_LDW(1, 1);                                                 \
    __builtin_amdgcn_sched_barrier(0);                            \
  }
#define G5_TAP_BODY(t, MF)                                        \
  {                                                               \
    constexpr int dx_ = (t) / 3, dy_ = (t) % 3;                   \
    MF(0, (t) & 1, dx_ & 1, dy_);                                 \
    G5_PIN(0);                                                    \
    __builtin_amdgcn_s_setprio(2);                                \
    __builtin_amdgcn_sched_barrier(0);                            \
    MF(1, (t) & 1, dx_ & 1, dy_);                                 \
    MF(2, (t) & 1, dx_ & 1, dy_);                                 \
    if constexpr (MT > 3) MF(3, (t) & 1, dx_ & 1, dy_);           \
    G5_PIN(1);                                                    \
    G5_PIN(2);                                                    \
    if constexpr (MT > 3) G5_PIN(3);                              \
    __builtin_amdgcn_sched_barrier(0);                            \
    __builtin_amdgcn_s_setprio(1);                                \
    __builtin_amdgcn_sched_barrier(0);                            \
  }
    constexpr int A_TAPS = (C::A_ITERS + 1) / 2;  // taps 0 .. A_TAPS-1 issue the patch pieces, two each
    static_assert(A_TAPS + C::B_ITERS - 1 <= 8, "DMA pieces are issued inside the 9 taps");

    // the lane's 16 bias values of this tile as four accumulator-shaped registers: read in every item (before everything
    // else: tap 0's wait covers them), used by the first tap of a tile's first item
    u32x4 bvec[4];
    {
      const int ba = C::BIAS_OFF + bslot * 256 + 64 * g;
      G5_RD(bvec[0], ba, 0);
      G5_RD(bvec[1], ba, 16);
      G5_RD(bvec[2], ba, 32);
      G5_RD(bvec[3], ba, 48);
    }
    G5_LDW(0, 0);
    G5_LDP(0, 0);
    __builtin_amdgcn_sched_barrier(0);

    // (MODE 2: an item ends with its chunk group)
    const bool last_chunk = MODE == 2 ? (chunk + 1) % a.grp == 0 : chunk + 1 == nchunks;
    const int nlid = last_chunk ? lid + nb : lid;
    const bool has_next = nlid < t_end;
    const Tile nxt = (last_chunk && has_next) ? decode(nlid) : cur;
    const int nchunk = last_chunk ? (MODE == 2 ? nxt.split * a.grp : 0) : chunk + 1;
    const char *nx_src = nullptr, *nw_src = nullptr;
    if (has_next) dma_bases(nxt, nchunk, nx_src, nw_src);
    const bool skip_b = resident && items_done >= 1;  // the NEXT item is this workgroup's third or later
    __builtin_amdgcn_sched_barrier(0);

    // in flight after each tap's wait (oldest first): see the header
    G5_TAP_HEAD0(G5_WAIT_WP)                 // [bias W0 P0]        -> all landed; then W1 requested
    asm volatile("" : "+v"(bvec[0]), "+v"(bvec[1]), "+v"(bvec[2]), "+v"(bvec[3]));
    if constexpr (MODE != 0) {
      // a chunk group starts from the bias (group 0) or from zero
      if (chunk != 0) bvec[0] = bvec[1] = bvec[2] = bvec[3] = u32x4{0u, 0u, 0u, 0u};
    }
    if (MODE != 0 ? chunk % a.grp == 0 : chunk == 0) {
      G5_TAP_BODY(0, G5_MFMA_B)
    } else {
      G5_TAP_BODY(0, G5_MFMA)
    }
    G5_TAP(1, G5_WAIT_W, 4, G5_LDP(1, 1))    // [W1 | W2]           -> W1; then P1 requested
    G5_TAP(2, G5_WAIT_W, R + 4, )            // [W2 | P1 W3]        -> W2
    G5_TAP(3, G5_WAIT_WP, 4, G5_LDP(2, 0))   // [P1 W3 | W4]        -> P1, W3; then P2 requested (set 0 is free)
    G5_TAP(4, G5_WAIT_W, R + 4, )            // [W4 | P2 W5]        -> W4
    G5_TAP(5, G5_WAIT_W, 4, )                // [P2 W5 | W6]        -> W5 (P2, older, has landed too)
    G5_TAP(6, G5_WAIT_WP, 4, )               // [W6 | W7]           -> W6; names P2's registers (set 0) before their first use
    G5_TAP(7, G5_WAIT_W, 4, )                // [W7 | W8]           -> W7
    G5_TAP(8, G5_WAIT_W, 0, )                // [W8]                -> W8

    if constexpr (MODE == 1) {
      if ((chunk + 1) % a.grp == 0) {       // a group is complete: out = ((g0 + g1) + g2) + ...  (plain fp32 adds, as the
#pragma unroll                              //  reduction kernel of the split form makes them)
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) tot[i][j] = chunk + 1 == a.grp ? acc[i][j] : tot[i][j] + acc[i][j];
      }
    }
    if constexpr (MODE == 2) {
      if (last_chunk) {
        // ---- item epilogue of the split form: the group's fp32 accumulators into this item's slab, 64 bytes per pixel and lane
        float* slab = a.slabs + (int64_t)lid * (C::TH * C::TW * 64);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          float* dst = slab + ((MT * wave + i) * C::TW + n15) * 64 + 16 * g;
#pragma unroll
          for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(dst + 4 * j) = acc[i][j];
        }
        in_flight = nstores;
        bslot ^= 1;
      }
    }
    if (MODE != 2 && last_chunk) {
      // ---- tile epilogue: bias, relu, bf16; two 16-byte stores per row and the fused 2x2 max-pool ----
      const int px = cur.x0 + n15;
      const int py = cur.y0 + MT * wave;
      const int cofs = cur.co0 + 16 * g;  // first of this lane's 16 consecutive channels
      char* const ymap = cur.second ? a.y2 : a.y;
      char* const qmap = cur.second ? a.pooled2 : a.pooled;
      // PACKED epilogue (as generation 4's 16x16 form): bias add in fp32, one v_cvt_pk_bf16_f32 per channel pair, relu and
      // the 2x2 max on the packed pairs as signed 16-bit integers -- 2-3 VALU instructions per value instead of ~10
      uint32_t pk[MT][8];
      // data-gradient extras (Conv2DBackpropInput + AddN + ReluGrad): every addend / mask word of the lane's MT pixels is
      // requested before the first is used
      u32x4 ad[MT][2], mk[MT][2];
      if constexpr (DG) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const int64_t off = (((int64_t)cur.n * (H + 2) + (py + i + 1)) * Wp + (px + 1)) * Ob + cofs * 2;
          if (a.addend != nullptr) {
            ad[i][0] = *reinterpret_cast<const u32x4*>(a.addend + off);
            ad[i][1] = *reinterpret_cast<const u32x4*>(a.addend + off + 16);
          }
          if (a.mask != nullptr) {
            mk[i][0] = *reinterpret_cast<const u32x4*>(a.mask + off);
            mk[i][1] = *reinterpret_cast<const u32x4*>(a.mask + off + 16);
          }
        }
      }
      // (the bias is already inside the accumulators: the tile's first MFMAs started from it; nothing to clear either)
      float sv[MT][16];
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) sv[i][r] = MODE == 1 ? tot[MODE == 1 ? i : 0][r >> 2][r & 3] : acc[i][r >> 2][r & 3];  // r = 4 j + q
      if constexpr (DG) {
        if (a.addend != nullptr) {  // added in fp32, before the one rounding (as every other generation)
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              const uint32_t w = k < 4 ? ad[i][0][k & 3] : ad[i][1][k & 3];
              sv[i][2 * k] += __builtin_bit_cast(float, w << 16);
              sv[i][2 * k + 1] += __builtin_bit_cast(float, w & 0xffff0000u);
            }
        }
      }
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int k = 0; k < 8; ++k) pk[i][k] = pack_bf16x2(sv[i][2 * k], sv[i][2 * k + 1]);  // channels 2 k, 2 k + 1 of the lane's 16
      if constexpr (DG) {
        if (a.mask != nullptr) {  // keep the value where the reference activation is > 0 (see conv_f8_dma.hip)
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              const uint32_t w = k < 4 ? mk[i][0][k & 3] : mk[i][1][k & 3];
              const uint32_t sel = ((int32_t)(w << 16) > 0 ? 0x0000ffffu : 0u) | ((int32_t)(w & 0xffff0000u) > 0 ? 0xffff0000u : 0u);
              pk[i][k] &= sel;
            }
        }
      }
      if (a.y != nullptr) {
        const uint32_t rfloor = a.relu ? 0u : 0x80008000u;  // (0x8000 = the smallest int16: a no-op)
#pragma unroll
        for (int i = 0; i < MT; ++i) {
#pragma unroll
          for (int k = 0; k < 8; ++k) pk[i][k] = pk_max_i16(pk[i][k], rfloor);
          char* dst = ymap + (((int64_t)cur.n * (H + 2) + (py + i + 1)) * Wp + (px + 1)) * Ob + cofs * 2;
          *reinterpret_cast<u32x4*>(dst) = u32x4{pk[i][0], pk[i][1], pk[i][2], pk[i][3]};
          *reinterpret_cast<u32x4*>(dst + 16) = u32x4{pk[i][4], pk[i][5], pk[i][6], pk[i][7]};
        }
      }
      if constexpr (MT % 2 == 0) {
        if (a.pooled != nullptr) {
          const int Hq = H >> 1, Wq = W >> 1;
#pragma unroll
          for (int i = 0; i < MT; i += 2) {
            uint32_t m[8];
            if (a.relu) {  // max over the 2x2 block, relu once on the result (see conv_f8_dma.hip)
#pragma unroll
              for (int k = 0; k < 8; ++k) {
                const uint32_t t = pk_max_i16(pk[i][k], pk[i + 1][k]);
                m[k] = pk_max_i16(pk_max_i16(t, pk_dpp_swap1(t)), 0u);
              }
            } else {
#pragma unroll
              for (int k = 0; k < 8; ++k) {
                const uint32_t t = pk_max_i16(pk_ord_bf16(pk[i][k]), pk_ord_bf16(pk[i + 1][k]));
                m[k] = pk_ord_bf16(pk_max_i16(t, pk_dpp_swap1(t)));
              }
            }
            // every wave issues these instructions (even lanes store): the counted vmcnt at the next barrier relies on it
            char* dst = qmap + (((int64_t)cur.n * (Hq + 2) + (((py + i) >> 1) + 1)) * (Wq + 2) + ((px >> 1) + 1)) * Ob + cofs * 2;
            if ((lane & 1) == 0) {
              *reinterpret_cast<u32x4*>(dst) = u32x4{m[0], m[1], m[2], m[3]};
              *reinterpret_cast<u32x4*>(dst + 16) = u32x4{m[4], m[5], m[6], m[7]};
            }
          }
        }
      }
      in_flight = nstores;
      bslot ^= 1;
    }
    if (!has_next) break;
    ++items_done;
    lid = nlid;
    cur = nxt;
    chunk = nchunk;
    buf ^= 1;
  }
#ifdef XV_CLOCK_STAMP
  XV_CLK_END(xv_clk_g5)
#endif
#undef G5_RD
#undef G5_LDW
#undef G5_LDP
#undef G5_WAIT_W
#undef G5_WAIT_WP
#undef G5_MFMA
#undef G5_MFMA_B
#undef G5_MFMA_C
#undef G5_C_ACC
#undef G5_C_BIAS
#undef G5_TAP_HEAD
#undef G5_TAP_HEAD0
#undef G5_TAP_BODY
#undef G5_PIN
#undef G5_DMA_PIECES
#undef G5_TAP
}

template <int MT, bool DG = false, int MODE = 0>
int g5_launch(const G5Args& a, int grid, hipStream_t stream) {
  static bool attr_set[XV_MAX_DEVICES] = {false};
  const hipError_t e = xv_allow_dynamic_lds(reinterpret_cast<const void*>(&conv_dma5_kernel<MT, DG, MODE>), G5<MT>::LDS_BYTES, attr_set);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL((conv_dma5_kernel<MT, DG, MODE>), dim3((unsigned)grid), dim3(G5<MT>::NT), G5<MT>::LDS_BYTES, stream, a);
  return xv_launch_status();
}

// The second half of the split form: out = relu(((g0 + g1) + g2) + ...) of a tile's slabs in group order (group 0 carries
// the bias), rounded to bf16 once, into the padded map.  One thread per pixel and 8 channels (two 16-byte reads per group,
// one 16-byte store).
template <int MT>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slabs, char* __restrict__ y, int N, int H,
                                                           int W, int Cout, int tiles_x, int tiles_y, int n_ct, int ngrp,
                                                           int relu) {
  constexpr int TH = 8 * MT, TW = 16, TPIX = TH * TW;
  const int64_t total = (int64_t)N * H * W * (Cout >> 3);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(i % (Cout >> 3));
    int64_t r = i / (Cout >> 3);
    const int px = (int)(r % W);
    r /= W;
    const int py = (int)(r % H);
    const int n = (int)(r / H);
    const int ty = py / TH, tx = px / TW, ct = c8 >> 3;
    const int64_t tile = (((int64_t)n * tiles_y + ty) * tiles_x + tx) * n_ct + ct;
    const float* src = slabs + (tile * ngrp) * (TPIX * 64) + ((py - ty * TH) * TW + (px - tx * TW)) * 64 + (c8 & 7) * 8;
    f32x4 a0 = *reinterpret_cast<const f32x4*>(src), a1 = *reinterpret_cast<const f32x4*>(src + 4);
    for (int s = 1; s < ngrp; ++s) {
      const float* p = src + (int64_t)s * (TPIX * 64);
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(p), b1 = *reinterpret_cast<const f32x4*>(p + 4);
      a0 = a0 + b0;
      a1 = a1 + b1;
    }
    u32x4 o = {pack_bf16x2(a0.x, a0.y), pack_bf16x2(a0.z, a0.w), pack_bf16x2(a1.x, a1.y), pack_bf16x2(a1.z, a1.w)};
    if (relu) o = u32x4{pk_max_i16(o.x, 0u), pk_max_i16(o.y, 0u), pk_max_i16(o.z, 0u), pk_max_i16(o.w, 0u)};
    *reinterpret_cast<u32x4*>(y + ((((int64_t)n * (H + 2) + py + 1) * (W + 2) + px + 1) * Cout + c8 * 8) * 2) = o;
  }
}

}  // namespace

// Does generation 5 take this shape with `mt` rows per wave?  (3x3, whole 32-channel chunks from 64 channels, maps that
// tile exactly in (8 mt) x 16)
bool xv_conv3x3_col_ok(int H, int W, int Cin, int Cout, int mt) {
  return (mt == 3 || mt == 4) && H > 0 && W > 0 && H % (8 * mt) == 0 && (W & 15) == 0 && Cin >= 64 && (Cin & 31) == 0 &&
         (Cout & 63) == 0;
}

// Chunks per group on this shape (= all of them: no grouping).  Groups of two from eight chunks on, on maps only generation 5
// serves -- a function of the layer shape alone, never of the batch size: every launch of the layer adds in the same order.
int xv_conv3x3_col_group(int H, int W, int Cin, int mt) {
  const int nchunks = Cin / 32;
  // XV_COL_GROUPS=<chunks per group> (0: no groups) for A/B timing; the default of four chunks keeps the fp32 slabs of the
  // split form (98 KB per item, written and read once) at half the bytes of two-chunk groups
  static const int env = getenv("XV_COL_GROUPS") != nullptr ? atoi(getenv("XV_COL_GROUPS")) : 4;
  if (env <= 0 || mt != 3 || ((H & 15) == 0 && (W & 31) == 0) || nchunks < 8 || (nchunks % env) != 0) return nchunks;
  return env;
}

// fp32 slabs of the split form: one per (tile, group); 0 where this shape / batch is not split (whole tiles fill at least
// half the CUs, or the shape has no chunk groups)
size_t xv_conv3x3_col_split_bytes(int N, int H, int W, int Cin, int Cout, int mt, int num_cus) {
  if (!xv_conv3x3_col_ok(H, W, Cin, Cout, mt)) return 0;
  const int grp = xv_conv3x3_col_group(H, W, Cin, mt), nchunks = Cin / 32;
  const int64_t ntiles = (int64_t)(W / 16) * (H / (8 * mt)) * N * (Cout / 64);
  if (grp == nchunks || 2 * ntiles >= num_cus) return 0;
  return (size_t)ntiles * (nchunks / grp) * (8 * mt * 16 * 64) * sizeof(float);
}

int xv_launch_conv3x3_col(const void* x, const void* wpk, const float* bias, void* y, void* pooled, int N, int H, int W, int Cin,
                          int Cout, int relu, int mt, int num_cus, hipStream_t stream, const void* mask, const void* addend,
                          void* split_ws, size_t split_bytes) {
  if (!xv_conv3x3_col_ok(H, W, Cin, Cout, mt) || (y == nullptr && pooled == nullptr) || (pooled != nullptr && (mt & 1)) ||
      ((mask != nullptr || addend != nullptr) && y == nullptr))
    return XV_ESHAPE;
  G5Args a{};
  a.x = (const char*)x;
  a.wpk = (const char*)wpk;
  a.bias = bias;
  a.y = (char*)y;
  a.pooled = (char*)pooled;
  a.N = N, a.H = H, a.W = W, a.Cin = Cin, a.Cout = Cout;
  a.tiles_x = W / 16;
  a.tiles_y = H / (8 * mt);
  a.n_ct = Cout / 64;
  const int64_t ntiles = (int64_t)a.tiles_x * a.tiles_y * N * a.n_ct;
  if (ntiles <= 0 || ntiles > 0x7fffffff) return XV_ESHAPE;
  a.n_tiles = (int)ntiles;
  a.relu = relu;
  a.mask = (const char*)mask;
  a.addend = (const char*)addend;
  const int grid = num_cus > 0 ? num_cus : 256;
  if (mask != nullptr || addend != nullptr) return mt == 3 ? g5_launch<3, true>(a, grid, stream) : g5_launch<4, true>(a, grid, stream);
  a.grp = xv_conv3x3_col_group(H, W, Cin, mt);
  a.ngrp = (Cin / 32) / a.grp;
  if (a.ngrp > 1) {       // (mt == 3 only)
    const size_t need = xv_conv3x3_col_split_bytes(N, H, W, Cin, Cout, mt, grid);
    if (need > 0 && split_ws != nullptr && split_bytes >= need && pooled == nullptr && y != nullptr &&
        ntiles * a.ngrp <= 0x7fffffff) {
      a.slabs = (float*)split_ws;
      a.n_tiles = (int)(ntiles * a.ngrp);
      const int rc = g5_launch<3, false, 2>(a, grid, stream);
      if (rc != XV_OK) return rc;
      const int64_t total = (int64_t)N * H * W * (Cout / 8);
      const int64_t blocks = (total + 255) / 256;
      hipLaunchKernelGGL(splitk_reduce_kernel<3>, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, stream,
                         (const float*)split_ws, (char*)y, N, H, W, Cout, a.tiles_x, a.tiles_y, a.n_ct, a.ngrp, relu);
      return xv_launch_status();
    }
    return g5_launch<3, false, 1>(a, grid, stream);
  }
  return mt == 3 ? g5_launch<3>(a, grid, stream) : g5_launch<4>(a, grid, stream);
}

// Two problems of one shape in ONE launch (G5Args::n_first); both full maps or neither, both pooled maps or neither.
int xv_launch_conv3x3_col_pair(const void* const x[2], const void* const wpk[2], const float* const bias[2], void* const y[2],
                               void* const pooled[2], int N, int H, int W, int Cin, int Cout, int relu, int mt, int num_cus,
                               hipStream_t stream) {
  if (!xv_conv3x3_col_ok(H, W, Cin, Cout, mt) || (y[0] == nullptr) != (y[1] == nullptr) ||
      (pooled[0] == nullptr) != (pooled[1] == nullptr) || (y[0] == nullptr && pooled[0] == nullptr) ||
      (pooled[0] != nullptr && (mt & 1)))
    return XV_ESHAPE;
  G5Args a{};
  a.x = (const char*)x[0], a.wpk = (const char*)wpk[0], a.bias = bias[0], a.y = (char*)y[0], a.pooled = (char*)pooled[0];
  a.x2 = (const char*)x[1], a.wpk2 = (const char*)wpk[1], a.bias2 = bias[1], a.y2 = (char*)y[1], a.pooled2 = (char*)pooled[1];
  a.n_first = N;
  a.N = 2 * N, a.H = H, a.W = W, a.Cin = Cin, a.Cout = Cout;
  a.tiles_x = W / 16;
  a.tiles_y = H / (8 * mt);
  a.n_ct = Cout / 64;
  const int64_t ntiles = (int64_t)a.tiles_x * a.tiles_y * a.N * a.n_ct;
  if (ntiles <= 0 || ntiles > 0x7fffffff) return XV_ESHAPE;
  a.n_tiles = (int)ntiles;
  a.relu = relu;
  const int grid = num_cus > 0 ? num_cus : 256;
  a.grp = xv_conv3x3_col_group(H, W, Cin, mt);
  a.ngrp = (Cin / 32) / a.grp;
  if (a.ngrp > 1) return g5_launch<3, false, 1>(a, grid, stream);      // the same chunk groups as a single problem
  return mt == 3 ? g5_launch<3>(a, grid, stream) : g5_launch<4>(a, grid, stream);
}

#ifdef XV_CLOCK_STAMP
extern "C" int xv_debug_read_clock_g5(void* dst, size_t bytes) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(xv_clk_g5), bytes); }
extern "C" int xv_debug_reset_clock_g5(void) {
  static unsigned long long zeros[4 * XV_CLK_SLOTS];
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(xv_clk_g5), zeros, sizeof(zeros));
}
#endif
