// Generation 4: the 3x3 convolution with every operand staged by LDS-DMA and one barrier per work item -- configuration 26
// on bf16 operands (v_mfma_f32_16x16x32_bf16; THE default 3x3 conv of the engine, forward and data gradient) and
// configuration 24 on e4m3 operands (v_mfma_scale_f32_32x32x64_f8f6f4; conv_dtype='fp8').
//
// Replaces, like the other conv kernels, tf.layers.conv2d(3x3, 'same', relu) + max_pooling2d of the FCN trunk
// (xview/models/simple_fcn.py:39-79) and, with the DG epilogue, the Conv2DBackpropInput + AddN + ReluGrad ops that
// tf.train.*Optimizer.minimize adds for them (base_model.py:153-162).
//
// Shared by both forms (the design of generation 2, conv_dma_kernel, with a leaner item loop):
//   * a work item = (16x32 pixel tile, 64 output channels, one 64-BYTE chunk of input channels: 32 bf16 or 64 e4m3): an
//     18x34 halo patch (39 KB) + all nine 64-row weight tiles (36 KB), both double buffered (151 KB of LDS), moved by
//     global_load_lds_dwordx4 written in assembly; the next item's pieces are issued two per tap, ONE barrier per item behind
//     a counted vmcnt that leaves the tile's own stores (younger than every DMA) in flight;
//   * 8 waves, wave w owns image rows 2w, 2w+1 x 32 columns x 64 channels (64 accumulator registers);
//   * LDS images: pixel / weight row r at r * 64 with its 16-byte slots swizzled so that every 16-lane group of a
//     ds_read_b128 covers the 64 banks once for all three horizontal taps (brute-forced per MFMA shape); an LDS-DMA writes
//     1 KB linearly, so the swizzle is applied on the SOURCE side (per-lane global offsets for the patch, pre-swizzled
//     packed images for the weights);
//   * the weight rows are permuted in the packed image so that a lane's 16 accumulator registers of a pixel are 16
//     CONSECUTIVE output channels (one 16- or 32-byte run per pixel and lane);
//   * dx-major taps, fragments one tap (weights) / half a column group (pixels) ahead in a second register set, issued and
//     waited for by hand (counted lgkmcnt, never more than 12 reads in flight), each tap's MFMAs as one burst at raised
//     priority; layers with one or two chunks keep their weights resident from the third item on;
//   * partial tiles (EDGE): clamped DMA source coordinates onto the zero border, predicated stores.
// bf16 form (M16): 16 MFMAs of K = 32 per tap on 4 x 4 blocks of 16x16 -- the chip holds ~0.2 GHz more clock on this shape
// than on 32x32x16 at equal cycles (profiles/r4_conv_inkernel_clock.json); the tile's first MFMAs start from the bias; packed
// bf16 epilogue (relu / 2x2 max on v_pk_max_i16), stores through a per-wave LDS stage so that four consecutive lanes write
// one 64-byte run; STATS: per-channel sums of the stored values for batch-norm training; DG: addend + relu mask.
// e4m3 form (F8): 4 block-scaled MFMAs of K = 64 per tap on 2 x 2 blocks of 32x32; first tap from C = 0; epilogue in two
// instructions per value (fma + v_med3_f32 = scale, bias, relu, saturation), pooling by v_med3_f32 against +inf.
#include "xv_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

struct F8Args {
  const char* x;      // e4m3 [N][H+2][W+2][Cin], zero border
  const char* wpk;    // packed fp8 weights: [256-byte header][generation-1 image][generation-4 image]
  const float* bias;  // [Cout]
  char* y;            // e4m3 [N][H+2][W+2][Cout] or null
  char* pooled;       // e4m3 [N][H/2+2][W/2+2][Cout] or null
  int N, H, W, Cin, Cout;
  int tiles_x, tiles_y, n_ct, n_tiles;
  int relu;
  int scale_x;    // E8M0 byte of the input map's scale in all four bytes
  float out_mul;  // 2^-scale_exp of the output map
  float* stats;   // STATS: [gridDim.x][2 Cout] per-workgroup sums / sums of squares of the stored outputs
  // data-gradient epilogue (bf16 16x16 form only): y = (conv + addend) where mask > 0, else 0 -- both maps shaped like y
  const char* mask;
  const char* addend;
  // SECOND PROBLEM of the same shape (the other expert of a fusion model: its own maps, weights and bias), n_first > 0:
  // images n_first .. N - 1 of the tile list are images 0 .. of these.  One launch then makes whole rounds of workgroups
  // where each expert alone leaves the last one half empty (conv4_x at 16 images: 2 x 1152 tiles = 9 rounds of 256
  // instead of 2 x 4.5 -> 2 x 5).  bf16 forward form only.
  const char* x2;
  const char* wpk2;
  const float* bias2;
  char* y2;
  char* pooled2;
  int n_first;
  // ROUTED POOL (RT kernels, training): one byte per POOLED value -- 0 = the window's maximum is not positive (no gradient
  // passes the relu), else 0x80 >> the first position of the maximum, in the order (row 0: columns 0, 1; row 1: columns 0, 1)
  // MaxPoolGrad visits.  The forward form writes [N][H/2][W/2][Cout] beside the pooled map INSTEAD of the full map; the
  // data-gradient form reads [N][H][W][Cout] and writes its result routed onto the map of twice the size (a.y).
  char* route;
};

#ifdef XV_CLOCK_STAMP
__device__ unsigned long long xv_clk_g4[4 * XV_CLK_SLOTS];
#endif

#ifdef XV_CONV_TRACE
// debug build only (`make trace`, tools/conv_trace4.py): per work item four cycle stamps of waves 0 and 4 (the two waves of
// one SIMD) of every 32nd workgroup, kept in the last spare 1.5 KB of LDS during the kernel (a global store per stamp would
// sit in the vmcnt queue this kernel counts on)
__device__ long long xv_trace_buf4[8 * 2 * 24 * 4];
#define G4_TRACE_LDS 1536
#define G4_STAMP(k)                                                                                       \
  if (!STATS && (blockIdx.x & 31) == 0 && trace_item < 24 && lane == 0 && (wave & 3) == 0)                \
    reinterpret_cast<long long*>(smem + C::LDS_BYTES_STAGE)[((wave >> 2) * 24 + trace_item) * 4 + (k)] = __builtin_readcyclecounter();
#else
#define G4_TRACE_LDS 0
#define G4_STAMP(k)
#endif

struct G4 {
  static constexpr int NWAVES = 8, NT = 512;
  static constexpr int TH = 16, TW = 32, HH = TH + 2, HW = TW + 2, NPIX = HH * HW;
  static constexpr int A_PIECES = (NPIX * 4 + 63) / 64;  // 1 KB per DMA wave-instruction
  static constexpr int A_BYTES = A_PIECES * 1024;
  static constexpr int B_PIECES = 9 * 4;  // 9 taps x (64 rows x 64 B)
  static constexpr int B_BYTES = B_PIECES * 1024;
  static constexpr int BIAS_OFF = 2 * (A_BYTES + B_BYTES);  // two 256-byte bias slots (tile parity)
  static constexpr int LDS_BYTES = BIAS_OFF + 512;
  static constexpr int STATS_OFF = LDS_BYTES;          // STATS: [wave 0..7][sum 64 | sum of squares 64] fp32
  static constexpr int LDS_BYTES_STATS = LDS_BYTES + 8 * 128 * 4;
  // STORE STAGE (bf16 16x16 form, packed epilogue): 1 KB per wave.  A lane holds 32 contiguous bytes of ONE pixel and the
  // 16 pixels of a block sit in consecutive lanes 128+ bytes apart, so a global_store_dwordx4 scattered 64 16-byte pieces
  // over 16 lines -- the address path takes four lanes per cycle only when they share a line: ~64 cycles per store
  // instruction, and with all 8 waves storing at once the tile's stores took 2 200-3 800 cycles.  Each 64-byte run of a
  // pixel goes through this stage instead and comes back with its four pieces in four CONSECUTIVE lanes.
  static constexpr int STAGE_OFF = LDS_BYTES;
  static constexpr int LDS_BYTES_STAGE = LDS_BYTES + NWAVES * 1024;
  static_assert(LDS_BYTES_STAGE + 1536 <= 160 * 1024, "does not fit the LDS");
  static constexpr int A_ITERS = (A_PIECES + NWAVES - 1) / NWAVES;
  static constexpr int B_ITERS = (B_PIECES + NWAVES - 1) / NWAVES;
  static constexpr int PROW = HW * 64;  // bytes between patch rows
  static_assert(LDS_BYTES_STATS <= 160 * 1024, "does not fit the LDS");
};

__device__ __forceinline__ int g4_swz(int row, int slot) { return slot ^ ((row >> 2) & 3); }
// the 16x16x32 form (M16): lane l reads row l & 15 (+ dx), logical slot l >> 4; slot s of row r sits at physical slot
// s ^ ((r >> 1) & 2) -- every 16-lane group of a ds_read_b128 then covers the 64 banks once for every column offset 0..18
// (tools/lds_swizzle_search.py --m16: brute force over the linear maps of the row bits)
__device__ __forceinline__ int g4_swz16(int row, int slot) { return slot ^ ((row >> 1) & 2); }
template <bool M16>
__device__ __forceinline__ int g4_swz_t(int row, int slot) {
  return M16 ? g4_swz16(row, slot) : g4_swz(row, slot);
}

// four fp32 -> four e4m3 bytes (round-to-nearest-even) of value * mul, saturating (as conv_mfma.hip pack_fp8x4)
__device__ __forceinline__ uint32_t g4_pack_fp8x4(float v0, float v1, float v2, float v3, float mul) {
  const float a = __builtin_amdgcn_fmed3f(v0 * mul, -448.f, 448.f);
  const float b = __builtin_amdgcn_fmed3f(v1 * mul, -448.f, 448.f);
  const float c = __builtin_amdgcn_fmed3f(v2 * mul, -448.f, 448.f);
  const float d = __builtin_amdgcn_fmed3f(v3 * mul, -448.f, 448.f);
  int p = 0;
  p = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, p, false);
  p = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, p, true);
  return (uint32_t)p;
}

// packed 16-bit integer instructions of the routed pool, named explicitly: written as vector arithmetic the compiler turns
// min(x ^ y, 1) into a compare and a select PER HALF (v_cmp_ne_u16 + SDWA twin + two v_cndmask + v_perm: 30 instructions per
// word where these are 13)
__device__ __forceinline__ uint32_t g4_pk_min_u16(uint32_t a, uint32_t k) {
  uint32_t d;
  asm("v_pk_min_u16 %0, %1, %2" : "=v"(d) : "v"(a), "s"(k));
  return d;
}
__device__ __forceinline__ uint32_t g4_pk_mul_lo_u16(uint32_t a, uint32_t b) {
  uint32_t d;
  asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ uint32_t g4_pk_lshrrev_b16(uint32_t sh, uint32_t k) {  // k >> sh per half
  uint32_t d;
  asm("v_pk_lshrrev_b16 %0, %1, %2" : "=v"(d) : "v"(sh), "s"(k));
  return d;
}
__device__ __forceinline__ uint32_t g4_pk_lshlrev_b16(uint32_t sh, uint32_t a) {  // a << sh per half
  uint32_t d;
  asm("v_pk_lshlrev_b16 %0, %1, %2" : "=v"(d) : "s"(sh), "v"(a));
  return d;
}
__device__ __forceinline__ uint32_t g4_pk_sign_i16(uint32_t a) {  // 0xffff where bit 15 of a half is set, else 0
  uint32_t d;
  asm("v_pk_ashrrev_i16 %0, %1, %2" : "=v"(d) : "s"(0x000f000f), "v"(a));
  return d;
}

__device__ __forceinline__ float g4_dpp_swap1(float v) {  // value of lane ^ 1 (quad_perm [1,0,3,2])
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
}

// F8: e4m3 input map and weights; OF8: e4m3 output maps (an e4m3 input implies them; <false, true> is the bf16 conv that
// writes the first e4m3 map of the fp8 graph)
// STATS (bf16 maps, training with batch norm): the epilogue also adds up, per output channel, the stored (bf16-rounded)
// outputs and their squares -- lanes by DPP row shifts, the workgroup's tiles in LDS, one row of a.stats per workgroup at the
// end (xv_bn_sums_from_rows adds the rows in a fixed tree): the statistics pass over the map (xv_bn_stats) disappears.
// EDGE: partial tiles possible (clamped DMA offsets, predicated stores); maps that tile exactly run the form without that
// code (it cost 3-5 % on the one- and two-chunk layers)
// M16 (bf16 operands only, configuration 26): the same item loop on v_mfma_f32_16x16x32_bf16 at the SAME output tile per
// wave (2 rows x 32 columns x 64 channels = 4 pixel blocks x 4 channel blocks of 16x16, again 64 accumulator registers): a
// tap is 16 MFMAs of K = 32 = the whole 32-channel chunk instead of 4 x 2 of K = 16 -- the same matrix-pipe cycles, the same
// ds_read_b128 count (4 weight + 8/3 pixel reads per tap) and the same fragment registers.  The chip holds a higher clock
// on this shape (MI355X_MICROARCH.md, DVFS give-back item 7: 1.12-1.15x the FLOP/s of the 32x32x16 loop on random data at
// equal cycles).  Lane l = (column n15 = l & 15, k-group / channel group g = l >> 4); weight rows permuted so that the 16
// accumulator registers of a lane and pixel block are channels 16 g .. 16 g + 15 (row 16 j + 4 g + q = channel 16 g + 4 j + q);
// fragment j (pixel half h) of a row pair sits 1 KB behind fragment 0, so one base register serves a whole set.
// DG (bf16 16x16 form only): the data-gradient epilogue -- y = (conv + addend) where mask > 0, else 0 -- as a kernel of its
// own, so that the forward kernel carries neither its registers nor its branches
// RT (bf16 16x16 form, exact tilings): the routed pool of training (F8Args::route).  Forward: pooled map + route bytes, no
// full map -- the layer's full-resolution output is read by nothing but MaxPoolGrad.  DG: MaxPoolGrad + ReluGrad in the
// epilogue -- the gradient of the pooled map never goes to memory, every value is stored to the window position its route
// byte names and zeros to the three others.
template <bool F8, bool OF8 = F8, bool STATS = false, bool EDGE = false, bool M16 = false, bool DG = false, bool RT = false>
__global__ __launch_bounds__(512, 2) void conv_dma4_kernel(F8Args a) {
  using C = G4;
  static_assert(!RT || (M16 && !OF8 && !STATS && !EDGE), "the routed pool exists in the packed epilogue, exact tilings");
  static_assert(M16 != F8, "bf16 operands run on the 16x16x32 form, e4m3 operands on the 32x32x64 one");
  static_assert(!DG || (M16 && !OF8 && !STATS), "the data-gradient epilogue exists in the packed epilogue only");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n31 = M16 ? (lane & 15) : (lane & 31), hh = M16 ? (lane >> 4) : (lane >> 5);  // M16: column n15, group g
  const int H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout;
  const int Wp = W + 2;
  const int Cb = F8 ? Cin : 2 * Cin, Ob = OF8 ? Cout : 2 * Cout;  // bytes per pixel of the input / output maps
  const int nchunks = Cb >> 6;                                   // 64 bytes of input channels per chunk
  int scale_w = 0;
  if constexpr (F8) {
    scale_w = ((127 + *reinterpret_cast<const int*>(a.wpk)) & 0xff) * 0x01010101;
    scale_w = __builtin_amdgcn_readfirstlane(scale_w);
  }
  // generation-4 image: fp8 -- behind the header and the generation-1 image; bf16 -- the third image of the packed buffer
  const char* const wimg = F8 ? a.wpk + 256 + (int64_t)9 * Cin * Cout : a.wpk + (int64_t)4 * 9 * Cin * Cout;
  const char* const wimg2 = a.wpk2 + (int64_t)4 * 9 * Cin * Cout;  // (second problem: bf16 only)

  // persistent workgroups, XCD-contiguous tile ranges, the output-channel tiles of one patch adjacent (as generation 2)
  const int G = gridDim.x, b = blockIdx.x;
  const int xcd = b & 7, bi = b >> 3;
  const int nb = (G - xcd + 7) >> 3;
  const int T = a.n_tiles;
  const int tq = T >> 3, trm = T & 7;
  const int t_begin = xcd * tq + (xcd < trm ? xcd : trm);
  const int t_end = t_begin + tq + (xcd < trm ? 1 : 0);
  int lid = t_begin + bi;
  if constexpr (STATS) {
    if (lid >= t_end) {  // no tile: a row of zeros
      for (int i = tid; i < 2 * Cout; i += C::NT) a.stats[(int64_t)b * 2 * Cout + i] = 0.f;
      return;
    }
  }
  if (lid >= t_end) return;

  struct Tile {
    int n, y0, x0, co0;
    int second;  // 1: a tile of the second problem (n counts its images).  (int: a bool member sent the struct through an
                 // alloca that the compiler promoted into LDS, on top of the patch buffer)
  };
  auto decode = [&](int l) {
    Tile t;
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
  // g = piece * 64 + lane holds physical slot g & 3 of pixel g >> 2
  int aoff[C::A_ITERS];
#pragma unroll
  for (int it = 0; it < C::A_ITERS; ++it) {
    const int g = (wave + it * C::NWAVES) * 64 + lane;
    int p = g >> 2;
    p = p < C::NPIX ? p : C::NPIX - 1;
    const int hy = p / C::HW, hx = p - hy * C::HW;
    aoff[it] = (hy * Wp + hx) * Cb + (g4_swz_t<M16>(hx, g & 3) << 4);
  }
  // LDS fragment addresses: pixel column n31 + dx of patch row 2 * wave, weight row n31 of a 32-row block.  The lane's two
  // 16-byte slots -- fp8: 2 hh and 2 hh + 1 (its 32-byte half of the K = 64 instruction); bf16: hh and 2 + hh (k-group hh
  // of the two K = 16 steps) -- sit at swizzled positions, so each has its own base
  // M16: ONE slot per lane (k-group g of the K = 32 step); the second fragment of a pair is the next 16 rows / columns,
  // 1 KB further on (same swizzle: bit 2 of the row does not change), so the "second base" is the first
  const int s0 = M16 ? hh : (F8 ? 2 * hh : hh), s1 = M16 ? hh : (F8 ? 2 * hh + 1 : 2 + hh);
  constexpr int F1 = M16 ? 1024 : 0;  // byte offset of a pair's second fragment behind its (second) base
  int pbase0[3], pbase1[3], wbase0, wbase1;
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) {
    const int hx = n31 + dx;
    pbase0[dx] = ((2 * wave) * C::HW + hx) * 64 + (g4_swz_t<M16>(hx, s0) << 4);
    pbase1[dx] = ((2 * wave) * C::HW + hx) * 64 + (g4_swz_t<M16>(hx, s1) << 4);
  }
  wbase0 = 2 * C::A_BYTES + n31 * 64 + (g4_swz_t<M16>(n31, s0) << 4);
  wbase1 = 2 * C::A_BYTES + n31 * 64 + (g4_swz_t<M16>(n31, s1) << 4);
  const int64_t tap_pitch = (int64_t)nchunks * Cout * 64;  // bytes between taps of the packed image

  // LDS-DMA in assembly (SGPR base + 32-bit VGPR offset; M0 = the wave's LDS destination), as generation 2
  auto dma16 = [&](const char* sbase, int voff, int lds_off) {
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_off), "v"(voff), "s"(sbase) : "memory");
  };
  auto dma_bases = [&](const Tile& t, int chunk, const char*& xsrc, const char*& wsrc) {
    xsrc = (t.second ? a.x2 : a.x) + (((int64_t)t.n * (H + 2) + t.y0) * Wp + t.x0) * Cb + chunk * 64;
    wsrc = (t.second ? wimg2 : wimg) + (((int64_t)chunk * Cout + t.co0) << 6);
  };
  // edge tiles (the patch reaches past the padded image): patch coordinates are clamped onto the zero border, i.e. the
  // source offset is recomputed with hy <= ylim, hx <= xlim (re-derived from the lane number inside this rare path, as in
  // generation 2); the LDS position (and its swizzle) is unchanged
  auto dma_a = [&](const char* xsrc, int it, int buf, bool edge, int ylim, int xlim) {
    const int piece = wave + it * C::NWAVES;
    if (piece < C::A_PIECES) {
      int voff = aoff[it];
      if (EDGE && edge) {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int g = piece * 64 + ln;
        int p = g >> 2;
        p = p < C::NPIX ? p : C::NPIX - 1;
        const int hy = p / C::HW, hx = p - hy * C::HW;
        voff = ((hy < ylim ? hy : ylim) * Wp + (hx < xlim ? hx : xlim)) * Cb + (g4_swz_t<M16>(hx, g & 3) << 4);
      }
      dma16(xsrc, voff, buf * C::A_BYTES + piece * 1024);
    }
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
  int chunk = 0, buf = 0, bslot = 0;
  {
    const char *xsrc, *wsrc;
    dma_bases(cur, 0, xsrc, wsrc);
#pragma unroll
    for (int it = 0; it < C::A_ITERS; ++it)
      dma_a(xsrc, it, 0, cur.y0 + C::TH > H || cur.x0 + C::TW > W, H + 1 - cur.y0, W + 1 - cur.x0);
#pragma unroll
    for (int it = 0; it < C::B_ITERS; ++it) dma_b(wsrc, it, 0);
    dma_bias(cur, true, 0);
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // M16: acc4[row i][channel pair j][pixel half h][jj] = the 16x16 block (row i, columns 16 h .., channel block 2 j + jj)
  f32x4 acc4[2][2][2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) acc4[i][j][h][jj] = f32x4{0.f, 0.f, 0.f, 0.f};

  // 16-byte store instructions per tile and wave
  float ssum[2][16], ssq[2][16];  // STATS only
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) ssum[j][r] = ssq[j][r] = 0.f;
  // RESIDENT WEIGHTS (as generation 2).  With one or two chunks per tile the item parity IS the chunk (or there is only one),
  // so weight buffer p only ever holds chunk p's weights -- of the same output-channel tile too when every workgroup of the
  // XCD group keeps its tile residue (nb % n_ct == 0).  From its third item on such a workgroup requests no weights at all:
  // half the DMA pieces of conv1_2 / conv2_1-like layers.
  const bool resident = nchunks <= 2 && (nb % a.n_ct) == 0 && a.n_first == 0;  // (two problems: two sets of weights)
  int items_done = 0;
  // (packed bf16 epilogue: 8 staged stores for the full map, ONE full-wave store per pixel half for the pooled map)
  constexpr bool STAGED = M16 && !OF8 && !STATS;
  // (routed pool: 2 more for the route bytes; its data-gradient form stores 2 rows x 4 window positions x 2 pixel halves,
  // each as two staged 64-byte rounds)
  const int nstores = RT ? (DG ? 32 : 4)
                      : STAGED ? (a.y != nullptr ? 8 : 0) + (a.pooled != nullptr ? 2 : 0)
                               : ((a.y != nullptr ? 4 : 0) + (a.pooled != nullptr ? 2 : 0)) * (OF8 ? 1 : 2);
  int in_flight = 0;  // stores issued after the last DMA of the previous item

  XV_CLK_BEGIN()
#ifdef XV_CONV_TRACE
  int trace_item = 0;
#endif
  while (true) {
    G4_STAMP(0)  // arrival at the item barrier
    // This item's operands have landed (each wave retires its own DMA; the tile stores issued after it may stay in
    // flight: vmcnt counts in issue order), and every wave has finished reading the other buffer pair.
    if (RT && in_flight == 32)
      asm volatile("s_waitcnt vmcnt(32)\n\ts_barrier" ::: "memory");
    else if (in_flight == 12)
      asm volatile("s_waitcnt vmcnt(12)\n\ts_barrier" ::: "memory");
    else if (in_flight == 10)
      asm volatile("s_waitcnt vmcnt(10)\n\ts_barrier" ::: "memory");
    else if (in_flight == 8)
      asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
    else if (in_flight == 6)
      asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");
    else if (in_flight == 4)
      asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
    else if (in_flight == 2)
      asm volatile("s_waitcnt vmcnt(2)\n\ts_barrier" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    in_flight = 0;
    G4_STAMP(1)  // barrier passed

    const int pb0[3] = {pbase0[0] + buf * C::A_BYTES, pbase0[1] + buf * C::A_BYTES, pbase0[2] + buf * C::A_BYTES};
    const int pb1[3] = {pbase1[0] + buf * C::A_BYTES, pbase1[1] + buf * C::A_BYTES, pbase1[2] + buf * C::A_BYTES};
    const int wb0 = wbase0 + buf * C::B_BYTES, wb1 = wbase1 + buf * C::B_BYTES;

    // Fragment registers: weights of tap t in set t & 1 ([channel block][slot]), the 4 patch rows of column group dx in
    // set dx & 1 ([row][slot]).  Issue order and the counted waits (reads return in order; never more than 12 in flight,
    // the counter holds 15): before the loop W0 Pa0; tap t: W(t+1) | wait | P part | MFMAs, the P parts being
    // Pb0, Pa1, Pb1, -, Pa2, Pb2, -, -, - (a = rows 0-1, b = rows 2-3 of the next column group's set).
    u32x4 wf[2][2][2], xf[2][4][2];
#define G4_RD(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
    // t = dx*3 + dy  ->  packed tap dy*3 + dx
#define G4_LDW(t, set)                                \
  {                                                   \
    constexpr int tap_ = (((t) % 3) * 3 + (t) / 3) * 4096; \
    G4_RD(wf[set][0][0], wb0, tap_);                  \
    G4_RD(wf[set][0][1], wb1, tap_ + F1);             \
    G4_RD(wf[set][1][0], wb0, tap_ + 2048);           \
    G4_RD(wf[set][1][1], wb1, tap_ + 2048 + F1);      \
  }
#define G4_LDPA(dx, set)                         \
  {                                              \
    G4_RD(xf[set][0][0], pb0[dx], 0);            \
    G4_RD(xf[set][0][1], pb1[dx], F1);           \
    G4_RD(xf[set][1][0], pb0[dx], C::PROW);      \
    G4_RD(xf[set][1][1], pb1[dx], C::PROW + F1); \
  }
#define G4_LDPB(dx, set)                         \
  {                                              \
    G4_RD(xf[set][2][0], pb0[dx], 2 * C::PROW);       \
    G4_RD(xf[set][2][1], pb1[dx], 2 * C::PROW + F1);  \
    G4_RD(xf[set][3][0], pb0[dx], 3 * C::PROW);       \
    G4_RD(xf[set][3][1], pb1[dx], 3 * C::PROW + F1);  \
  }
    // at most n newer reads outstanding: wf[ws] (and rows 0-1 / rows 2-3 of xf[ps]) have landed.  Every wait names exactly
    // the registers it releases: the MFMAs that consume them cannot move above it, and no register with a read still in
    // flight is an operand of anything.
#define G4_WAIT_W(n, ws, ps) \
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(wf[ws][0][0]), "+v"(wf[ws][0][1]), "+v"(wf[ws][1][0]), "+v"(wf[ws][1][1]) : "n"(n))
#define G4_WAIT_WA(n, ws, ps)                                                                                         \
  asm volatile("s_waitcnt lgkmcnt(%8)"                                                                                \
               : "+v"(wf[ws][0][0]), "+v"(wf[ws][0][1]), "+v"(wf[ws][1][0]), "+v"(wf[ws][1][1]), "+v"(xf[ps][0][0]),  \
                 "+v"(xf[ps][0][1]), "+v"(xf[ps][1][0]), "+v"(xf[ps][1][1])                                           \
               : "n"(n))
#define G4_WAIT_WB(n, ws, ps)                                                                                         \
  asm volatile("s_waitcnt lgkmcnt(%8)"                                                                                \
               : "+v"(wf[ws][0][0]), "+v"(wf[ws][0][1]), "+v"(wf[ws][1][0]), "+v"(wf[ws][1][1]), "+v"(xf[ps][2][0]),  \
                 "+v"(xf[ps][2][1]), "+v"(xf[ps][3][0]), "+v"(xf[ps][3][1])                                           \
               : "n"(n))
#define G4_CAT(lo, hi) i32x8{(int)(lo).x, (int)(lo).y, (int)(lo).z, (int)(lo).w, (int)(hi).x, (int)(hi).y, (int)(hi).z, (int)(hi).w}
#define G4_C_ACC(i, j, h, jj) acc4[i][j][h][jj]
    // first tap of a tile (M16): the accumulation STARTS from the bias (lane's channels 16 g + 4 (2 j + jj) + q) -- no bias add and
    // no accumulator clearing in the epilogue (128 of its ~230 VALU instructions per lane)
#define G4_C_BIAS(i, j, h, jj) __builtin_bit_cast(f32x4, bvec[2 * (j) + (jj)])
#define G4_MFMA(i, j, ws, ps, dy) G4_MFMA_C(i, j, ws, ps, dy, G4_C_ACC)
    // e4m3 form, first tap of a tile: C = 0 (an inline constant: no register, no clearing in the epilogue)
#define G4_MFMA_Z(i, j, ws, ps, dy)                                                                                    \
  acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(G4_CAT(wf[ws][j][0], wf[ws][j][1]),                      \
                                                              G4_CAT(xf[ps][(i) + (dy)][0], xf[ps][(i) + (dy)][1]),    \
                                                              f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, \
                                                              0, 0, 0, scale_w, 0, a.scale_x);
#define G4_MFMA_B(i, j, ws, ps, dy) G4_MFMA_C(i, j, ws, ps, dy, G4_C_BIAS)
#define G4_MFMA_C(i, j, ws, ps, dy, CS)                                                                                \
  if constexpr (F8) {                                                                                                  \
    acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(G4_CAT(wf[ws][j][0], wf[ws][j][1]),                    \
                                                                G4_CAT(xf[ps][(i) + (dy)][0], xf[ps][(i) + (dy)][1]),  \
                                                                acc[i][j], 0, 0, 0, scale_w, 0, a.scale_x);            \
  } else if constexpr (M16) {                                                                                          \
    /* wf[ws][j][jj]: channel block 2 j + jj; xf[ps][row][h]: pixel half h */                                          \
    acc4[i][j][0][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[ws][j][0]),               \
                                                               __builtin_bit_cast(bf16x8, xf[ps][(i) + (dy)][0]), CS(i, j, 0, 0), 0, 0, 0); \
    acc4[i][j][0][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[ws][j][1]),               \
                                                               __builtin_bit_cast(bf16x8, xf[ps][(i) + (dy)][0]), CS(i, j, 0, 1), 0, 0, 0); \
    acc4[i][j][1][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[ws][j][0]),               \
                                                               __builtin_bit_cast(bf16x8, xf[ps][(i) + (dy)][1]), CS(i, j, 1, 0), 0, 0, 0); \
    acc4[i][j][1][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[ws][j][1]),               \
                                                               __builtin_bit_cast(bf16x8, xf[ps][(i) + (dy)][1]), CS(i, j, 1, 1), 0, 0, 0); \
  } else {                                                                                                             \
    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[ws][j][0]),                      \
                                                        __builtin_bit_cast(bf16x8, xf[ps][(i) + (dy)][0]), acc[i][j], 0, 0, 0); \
    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[ws][j][1]),                      \
                                                        __builtin_bit_cast(bf16x8, xf[ps][(i) + (dy)][1]), acc[i][j], 0, 0, 0); \
  }
    // the next item's DMA: patch pieces two per tap from tap 0 (HBM / Infinity-Cache latency), weight pieces (L2) after,
    // the bias with the first weight piece
#define G4_DMA_PIECES(t)                                                                 \
  if (has_next) {                                                                        \
    if (2 * (t) < C::A_ITERS) dma_a(nx_src, 2 * (t), buf ^ 1, nx_edge, nx_ylim, nx_xlim);         \
    if (2 * (t) + 1 < C::A_ITERS) dma_a(nx_src, 2 * (t) + 1, buf ^ 1, nx_edge, nx_ylim, nx_xlim); \
    if ((t) >= A_TAPS && (t) - A_TAPS < C::B_ITERS && !skip_b) dma_b(nw_src, (t) - A_TAPS, buf ^ 1); \
    if ((t) == A_TAPS) dma_bias(nxt, last_chunk, bslot ^ 1);                             \
  }
    // An MFMA is a pure value to the optimizer: nothing orders it against the (volatile) asm reads, waits and priority
    // changes around it, and left alone the MFMAs of several taps sink into one cluster behind them (seen in the ISA: the
    // priority pairs back to back with nothing between).  An empty volatile asm that "modifies" an accumulator pins the
    // MFMA that produced it in front of every later asm statement.
#define G4_PIN(i, j)                                                                                                  \
  if constexpr (M16)                                                                                                  \
    asm volatile("" : "+v"(acc4[i][j][0][0]), "+v"(acc4[i][j][0][1]), "+v"(acc4[i][j][1][0]), "+v"(acc4[i][j][1][1])); \
  else                                                                                                                \
    asm volatile("" : "+v"(acc[i][j]))
#define G4_TAP(t, WAIT, NEWER, POST) \
  {                                  \
    G4_TAP_HEAD(t, WAIT, NEWER, POST) \
    G4_TAP_BODY(t, G4_MFMA)          \
  }
#define G4_TAP_HEAD(t, WAIT, NEWER, POST)                         \
  {                                                               \
    if constexpr ((t) + 1 < 9) G4_LDW((t) + 1, ((t) + 1) & 1);    \
    G4_DMA_PIECES(t)                                              \
    WAIT(NEWER, (t) & 1, ((t) / 3) & 1);                          \
    POST;                                                         \
    __builtin_amdgcn_sched_barrier(0);                            \
  }
    // tap 0: the item starts with bias (M16) + W0 + Pa0 = 12 reads in flight; W1 is requested only behind the wait for
    // them (with it in front the counter would have to hold 16: it holds 15), and has all of tap 0's MFMAs to land
#define G4_TAP_HEAD0(WAIT, POST)                                  \
  {                                                               \
    G4_DMA_PIECES(0)                                              \
    WAIT(0, 0, 0);                                                \
    G4_LDW(1, 1);                                                 \
    POST;                                                         \
    __builtin_amdgcn_sched_barrier(0);                            \
  }
#define G4_TAP_BODY(t, MF)                                        \
  {                                                               \
    constexpr int dx_ = (t) / 3, dy_ = (t) % 3;                   \
    MF(0, 0, (t) & 1, dx_ & 1, dy_);                              \
    G4_PIN(0, 0);                                                 \
    __builtin_amdgcn_s_setprio(2);                                \
    __builtin_amdgcn_sched_barrier(0);                            \
    MF(0, 1, (t) & 1, dx_ & 1, dy_);                              \
    MF(1, 0, (t) & 1, dx_ & 1, dy_);                              \
    MF(1, 1, (t) & 1, dx_ & 1, dy_);                              \
    G4_PIN(0, 1);                                                 \
    G4_PIN(1, 0);                                                 \
    G4_PIN(1, 1);                                                 \
    __builtin_amdgcn_sched_barrier(0);                            \
    __builtin_amdgcn_s_setprio(1);                                \
    __builtin_amdgcn_sched_barrier(0);                            \
  }
    constexpr int A_TAPS = (C::A_ITERS + 1) / 2;  // taps 0 .. A_TAPS-1 issue the patch pieces, two each
    static_assert(A_TAPS + C::B_ITERS - 1 <= 8, "DMA pieces are issued inside the 9 taps");

    // (M16) the lane's 16 bias values of this tile (its slot landed with the tile's first operands) as four accumulator-shaped
    // registers, requested before everything else so that tap 0's wait covers them.  Read in EVERY item, used in a tile's
    // first: only the 16 MFMAs of tap 0 differ between the two cases (a branch around the reads too made the two paths
    // allocate their fragment registers differently: 16 two-register copies per item).
    u32x4 bvec[4];
    if constexpr (M16) {
      const int ba = C::BIAS_OFF + bslot * 256 + 64 * hh;
      G4_RD(bvec[0], ba, 0);
      G4_RD(bvec[1], ba, 16);
      G4_RD(bvec[2], ba, 32);
      G4_RD(bvec[3], ba, 48);
    }
    G4_LDW(0, 0);
    G4_LDPA(0, 0);
    __builtin_amdgcn_sched_barrier(0);

    const bool last_chunk = chunk + 1 == nchunks;
    const int nlid = last_chunk ? lid + nb : lid;
    const bool has_next = nlid < t_end;
    const Tile nxt = (last_chunk && has_next) ? decode(nlid) : cur;
    const int nchunk = last_chunk ? 0 : chunk + 1;
    const char *nx_src = nullptr, *nw_src = nullptr;
    if (has_next) dma_bases(nxt, nchunk, nx_src, nw_src);
    const bool nx_edge = nxt.y0 + C::TH > H || nxt.x0 + C::TW > W;
    const int nx_ylim = H + 1 - nxt.y0, nx_xlim = W + 1 - nxt.x0;
    const bool skip_b = resident && items_done >= 1;  // the NEXT item is this workgroup's third or later
    __builtin_amdgcn_sched_barrier(0);

    G4_TAP_HEAD0(G4_WAIT_WA, G4_LDPB(0, 0))
    if constexpr (M16) {
      // (the bias reads are older than W0: the wait that released W0 has released them; name them for the optimizer)
      asm volatile("" : "+v"(bvec[0]), "+v"(bvec[1]), "+v"(bvec[2]), "+v"(bvec[3]));
      if (chunk == 0) {
        G4_TAP_BODY(0, G4_MFMA_B)
      } else {
        G4_TAP_BODY(0, G4_MFMA)
      }
    } else {
      if (chunk == 0) {
        G4_TAP_BODY(0, G4_MFMA_Z)
      } else {
        G4_TAP_BODY(0, G4_MFMA)
      }
    }
    G4_TAP(1, G4_WAIT_WB, 4, G4_LDPA(1, 1))
    G4_TAP(2, G4_WAIT_W, 8, G4_LDPB(1, 1))
    G4_TAP(3, G4_WAIT_WA, 8, )
    G4_TAP(4, G4_WAIT_WB, 4, G4_LDPA(2, 0))
    G4_TAP(5, G4_WAIT_W, 8, G4_LDPB(2, 0))
    G4_TAP(6, G4_WAIT_WA, 8, )
    G4_TAP(7, G4_WAIT_WB, 4, )
    G4_TAP(8, G4_WAIT_W, 0, )
    G4_STAMP(2)  // all MFMAs of the item issued

    if (last_chunk) {
      // ---- tile epilogue: bias, relu, e4m3; one 16-byte store per (row, channel block) and the fused 2x2 max-pool ----
      const float* bl = reinterpret_cast<const float*>(smem + C::BIAS_OFF + bslot * 256);
      const int py = cur.y0 + 2 * wave;
      // 32x32 form: u = channel block j (the lane's 16 channels 32 j + 16 hh .., one pixel column n31);
      // 16x16 form: u = pixel half h (column 16 h + n15, the lane's 16 channels 16 g ..)
      // Data gradient: the addend / relu-reference words of a pixel half (2 rows x 16 pixels x 128 B x 2 maps) are requested
      // in ONE batch before anything uses them, and COALESCED: lane l asks for piece l & 3 of pixel l >> 2 of a 64-byte half
      // -- four consecutive lanes share a line -- and the pieces reach the lanes that own them (pixel n15, channel group g)
      // through the wave's store stage.  The second half's batch is requested as soon as the first has left its landing
      // registers, so it travels while the first half is packed and stored.  (16-byte loads at a one-pixel lane stride, one
      // round trip per row pair: the epilogue took 16-18 thousand cycles per tile, tools/conv_trace4.py --dgrad.)
      u32x4 ldA[2][2], ldM[2][2];  // [row i][64-byte half]
      auto dg_request = [&](int u) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int pxl = cur.x0 + 16 * u + (lane >> 2);
          const bool in = !EDGE || (py + i < H && pxl < W);
          const int64_t off = (((int64_t)cur.n * (H + 2) + ((in ? py + i : 0) + 1)) * Wp + ((in ? pxl : 0) + 1)) * Ob + cur.co0 * 2 +
                              (lane & 3) * 16;
#pragma unroll
          for (int half = 0; half < 2; ++half) {
            if (a.addend != nullptr) ldA[i][half] = *reinterpret_cast<const u32x4*>(a.addend + off + half * 64);
            if (a.mask != nullptr) ldM[i][half] = *reinterpret_cast<const u32x4*>(a.mask + off + half * 64);
          }
        }
      };
      if constexpr (DG && !RT) dg_request(0);
      // routed pool, data gradient: the route bytes of the lane's 16 channels at its 2 rows x 2 pixel halves, requested at once
      u32x4 rt[2][2];
      if constexpr (DG && RT) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int i = 0; i < 2; ++i)
            rt[u][i] = *reinterpret_cast<const u32x4*>(a.route + (((int64_t)cur.n * H + (py + i)) * W + (cur.x0 + 16 * u + n31)) * Cout +
                                                       cur.co0 + 16 * hh);
      }
      // a lane's 16 channels of one pixel (8 packed words) to `dst` through the wave's store stage (see the full-map stores)
      auto staged_store = [&](const uint32_t (&w8)[8], char* dst) {
        u32x4* const stage = reinterpret_cast<u32x4*>(smem + C::STAGE_OFF + wave * 1024);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          if ((hh >> 1) == half) {
            stage[n31 * 4 + 2 * (hh & 1)] = u32x4{w8[0], w8[1], w8[2], w8[3]};
            stage[n31 * 4 + 2 * (hh & 1) + 1] = u32x4{w8[4], w8[5], w8[6], w8[7]};
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          const u32x4 r = stage[lane];
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          *reinterpret_cast<u32x4*>(dst + half * 64) = r;
        }
      };
      char* const ymap = cur.second ? a.y2 : a.y;
      char* const qmap = cur.second ? a.pooled2 : a.pooled;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int cl = M16 ? 16 * hh : 32 * u + 16 * hh;  // first of this lane's 16 consecutive channels within the tile
        const int px = cur.x0 + (M16 ? 16 * u + n31 : n31);
        float bv[16];
        if constexpr (!M16) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 t4 = *reinterpret_cast<const f32x4*>(bl + cl + 4 * q);
            bv[4 * q] = t4.x, bv[4 * q + 1] = t4.y, bv[4 * q + 2] = t4.z, bv[4 * q + 3] = t4.w;
          }
        }
        if constexpr (M16 && !OF8) {
          // PACKED epilogue of the bf16 16x16 form.  The float form below spends ~10 VALU instructions per output value
          // (fmaxf under IEEE mode quiets both operands; relu, vertical and horizontal pool each pay it): 3 000-3 900 cycles
          // per wave and tile, exposed -- every wave is in its epilogue at once -- and a fifth of a two-chunk layer's time
          // (tools/conv_trace4.py).  Here: bias add in fp32, ONE v_cvt_pk_bf16_f32 per channel pair, then relu and the 2x2 max
          // on the packed pairs as signed 16-bit integers (a non-negative bf16 orders like its bit pattern, rounding is
          // monotone, so pooling after rounding equals rounding after pooling): 2-3 instructions per value.
          uint32_t pk[2][8];
          // data-gradient extras (Conv2DBackpropInput + AddN + ReluGrad): the words requested above, through the stage to the
          // lane that owns the pixel and channel group (a lane's 32 bytes are two pieces of ONE 64-byte half: it reads them in
          // the round of its half); then the other pixel half's request
          u32x4 ad[2][2], mk[2][2];
          if constexpr (DG && !RT) {
            u32x4* const xstage = reinterpret_cast<u32x4*>(smem + C::STAGE_OFF + wave * 1024);
            auto to_owner = [&](const u32x4 (&ld)[2], u32x4 (&own)[2]) {
#pragma unroll
              for (int half = 0; half < 2; ++half) {
                xstage[lane] = ld[half];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if ((hh >> 1) == half) {
                  own[0] = xstage[n31 * 4 + 2 * (hh & 1)];
                  own[1] = xstage[n31 * 4 + 2 * (hh & 1) + 1];
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
              }
            };
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              if (a.addend != nullptr) to_owner(ldA[i], ad[i]);
              if (a.mask != nullptr) to_owner(ldM[i], mk[i]);
            }
            if (u == 0) dg_request(1);
          }
          // (the bias is already inside the accumulators: the tile's first MFMAs started from it; nothing to clear either)
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            float sv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) sv[r] = acc4[i][r >> 3][u][(r >> 2) & 1][r & 3];  // r = 4 (2 j + jj) + q
            if constexpr (DG && !RT) {
              if (a.addend != nullptr) {  // added in fp32, before the one rounding (as every other generation)
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                  const uint32_t w = k < 4 ? ad[i][0][k & 3] : ad[i][1][k & 3];
                  sv[2 * k] += __builtin_bit_cast(float, w << 16);
                  sv[2 * k + 1] += __builtin_bit_cast(float, w & 0xffff0000u);
                }
              }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) pk[i][k] = pack_bf16x2(sv[2 * k], sv[2 * k + 1]);  // channels 2 k, 2 k + 1 of the lane's 16
            if constexpr (DG && !RT) {
              if (a.mask != nullptr) {
                // keep the value where the reference activation is > 0: each bf16 half moved to the top of a 32-bit integer
                // (negative values, -0 and +0 are <= 0 there).  (A packed 16-bit min / max form of this was miscompiled by
                // hipcc 7.2: every pair tested the mask word of pair 0.)
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                  const uint32_t w = k < 4 ? mk[i][0][k & 3] : mk[i][1][k & 3];
                  const uint32_t sel = ((int32_t)(w << 16) > 0 ? 0x0000ffffu : 0u) | ((int32_t)(w & 0xffff0000u) > 0 ? 0xffff0000u : 0u);
                  pk[i][k] &= sel;
                }
              }
            }
          }
          if constexpr (DG && RT) {
            // MaxPoolGrad + ReluGrad: window position q of pooled pixel (py + i, pxl) is pixel (2 (py + i) + (q >> 1),
            // 2 pxl + (q & 1)) of the full map; it takes the value where the route byte is 0x80 >> q, zero elsewhere (all
            // four positions are written: the map needs no clearing).
            const int pxl = cur.x0 + 16 * u + (lane >> 2);
            const int64_t Wf = 2 * (int64_t)W + 2;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              uint32_t cd[8];
#pragma unroll
              for (int k = 0; k < 8; ++k)  // bytes 2 k, 2 k + 1 of the lane's 16 -> the HIGH bytes of a word's two halves
                cd[k] = __builtin_amdgcn_perm(0u, rt[u][i][k >> 1], (k & 1) ? 0x030c020cu : 0x010c000cu);
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                uint32_t w8[8];
#pragma unroll
                for (int k = 0; k < 8; ++k)  // position q's bit (0x80 >> q of the byte) to the sign, the sign over the half
                  w8[k] = pk[i][k] & g4_pk_sign_i16(q == 0 ? cd[k] : g4_pk_lshlrev_b16(q * 0x00010001u, cd[k]));
                staged_store(w8, a.y + (((int64_t)cur.n * (2 * H + 2) + (2 * (py + i) + (q >> 1) + 1)) * Wf + (2 * pxl + (q & 1) + 1)) * Ob +
                                     cur.co0 * 2 + (lane & 3) * 16);
              }
            }
            continue;
          }
          if (a.y != nullptr) {
            const uint32_t rfloor = a.relu ? 0u : 0x80008000u;  // (0x8000 = the smallest int16: a no-op)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
              for (int k = 0; k < 8; ++k) pk[i][k] = pk_max_i16(pk[i][k], rfloor);
              if constexpr (STATS) {
                // per lane and channel: sums of the STORED (bf16) values and of their squares, kept in registers over all the
                // workgroup's tiles (every tile of a workgroup has the same output-channel tile: the launcher checks
                // nb % n_ct == 0); both pixel halves are the same 16 channels; lanes, waves and workgroups meet after the last
                // tile.  The statistics kernel writes its rows straight from the lanes (no store stage: that LDS holds the
                // waves' sums at the end).
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                  const float a0 = __builtin_bit_cast(float, pk[i][k] << 16), a1 = __builtin_bit_cast(float, pk[i][k] & 0xffff0000u);
                  ssum[0][2 * k] += a0;
                  ssum[0][2 * k + 1] += a1;
                  ssq[0][2 * k] = fmaf(a0, a0, ssq[0][2 * k]);
                  ssq[0][2 * k + 1] = fmaf(a1, a1, ssq[0][2 * k + 1]);
                }
                char* dst = ymap + (((int64_t)cur.n * (H + 2) + (py + i + 1)) * Wp + (px + 1)) * Ob + (cur.co0 + cl) * 2;
                *reinterpret_cast<u32x4*>(dst) = u32x4{pk[i][0], pk[i][1], pk[i][2], pk[i][3]};
                *reinterpret_cast<u32x4*>(dst + 16) = u32x4{pk[i][4], pk[i][5], pk[i][6], pk[i][7]};
                continue;
              }
              // through the wave's store stage: [pixel 0..15][piece 0..3] = one 64-byte run per pixel; lanes of channel
              // groups 0-1 fill it for the pixel's bytes 0..63, then groups 2-3 for bytes 64..127; lane l reads back piece
              // l & 3 of pixel l >> 2 (LDS operations of one wave execute in order: no wait between the rounds)
              u32x4* stage = reinterpret_cast<u32x4*>(smem + C::STAGE_OFF + wave * 1024);
              const int pxl = cur.x0 + 16 * u + (lane >> 2);
              char* dst = ymap + (((int64_t)cur.n * (H + 2) + (py + i + 1)) * Wp + (pxl + 1)) * Ob + cur.co0 * 2 + (lane & 3) * 16;
              const bool on = !EDGE || (py + i < H && pxl < W);
#pragma unroll
              for (int half = 0; half < 2; ++half) {
                if ((hh >> 1) == half) {
                  stage[n31 * 4 + 2 * (hh & 1)] = u32x4{pk[i][0], pk[i][1], pk[i][2], pk[i][3]};
                  stage[n31 * 4 + 2 * (hh & 1) + 1] = u32x4{pk[i][4], pk[i][5], pk[i][6], pk[i][7]};
                }
                // lanes exchange data through LDS: a wave-scope fence on both sides (without it the compiler keeps a lane's
                // previous read where the lane itself has not written -- per-thread reasoning; seen in the ISA)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const u32x4 r = stage[lane];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (on) *reinterpret_cast<u32x4*>(dst + half * 64) = r;
              }
            }
          }
          if (a.pooled != nullptr) {
            uint32_t m[8];
            if (a.relu) {
              // max over the 2x2 block first, relu once on the result: with at least one positive value the integer order is
              // the float order among the candidates that can win, with none the result is clamped to zero anyway
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
            // the 8 pooled pixels of this half (even lanes) through the stage as [64-byte half 0..1][pixel 0..7][piece 0..3]:
            // ONE full-wave store -- lanes 0-31 the pixels' bytes 0..63, lanes 32-63 their bytes 64..127
            u32x4* stage = reinterpret_cast<u32x4*>(smem + C::STAGE_OFF + wave * 1024);
            if ((lane & 1) == 0) {
              const int at = ((hh >> 1) * 8 + (n31 >> 1)) * 4 + 2 * (hh & 1);
              stage[at] = u32x4{m[0], m[1], m[2], m[3]};
              stage[at + 1] = u32x4{m[4], m[5], m[6], m[7]};
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const u32x4 r = stage[lane];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int qx = ((cur.x0 + 16 * u) >> 1) + ((lane & 31) >> 2);
            char* dst = qmap + (((int64_t)cur.n * (Hq + 2) + ((py >> 1) + 1)) * (Wq + 2) + (qx + 1)) * Ob + cur.co0 * 2 +
                        (lane >> 5) * 64 + (lane & 3) * 16;
            // every wave issues this instruction: the counted vmcnt at the next barrier relies on it
            if (!EDGE || (py < H && 2 * qx < W)) *reinterpret_cast<u32x4*>(dst) = r;
            if constexpr (RT && !DG) {
              // route bytes (relu form; the launcher takes no other): per channel 0x80 >> the first window position whose value
              // IS the maximum -- (row 0, own column), (row 0, lane ^ 1's column), (row 1, own), else (row 1, the other) -- and 0
              // where the clamped maximum is 0.  Packed 16-bit arithmetic on the words the pool used: ne = min(v ^ max, 1) is 0
              // where a value equals the maximum; equal bits are equal values here, the maximum being positive wherever the
              // code is kept.  Even lanes (own column = window column 0) store, 16 bytes = the lane's 16 channels.
              constexpr uint32_t K1 = 0x00010001u;
              uint32_t rc[8];
#pragma unroll
              for (int k = 0; k < 8; ++k) {
                const uint32_t ne0 = g4_pk_min_u16(pk[0][k] ^ m[k], K1);
                const uint32_t ne1 = g4_pk_min_u16(pk_dpp_swap1(pk[0][k]) ^ m[k], K1);
                const uint32_t ne2 = g4_pk_min_u16(pk[1][k] ^ m[k], K1);
                const uint32_t n01 = ne0 & ne1;
                // 0x80 >> (position of the first maximum), times (maximum != 0)   (no carries between the halves: sums <= 3)
                rc[k] = g4_pk_mul_lo_u16(g4_pk_lshrrev_b16(ne0 + n01 + (n01 & ne2), 0x00800080u), g4_pk_min_u16(m[k], K1));
              }
              u32x4 rb;
#pragma unroll
              for (int j = 0; j < 4; ++j) rb[j] = __builtin_amdgcn_perm(rc[2 * j + 1], rc[2 * j], 0x06040200u);
              const int rx = (cur.x0 + 16 * u + n31) >> 1;
              // (every wave issues this instruction, as above)
              if ((lane & 1) == 0)
                *reinterpret_cast<u32x4*>(a.route + (((int64_t)cur.n * Hq + (py >> 1)) * Wq + rx) * Cout + cur.co0 + 16 * hh) = rb;
            }
          }
          continue;
        }
        if constexpr (OF8 && !STATS) {
          // e4m3 epilogue.  Scale (a power of two), bias, relu and saturation in TWO instructions per value: fma(s, mul, b mul)
          // = (s + b) mul exactly (scaling by a power of two commutes with rounding), v_med3_f32 against (0 | -448, 448) is
          // relu + clamp; the 2x2 max on those (scaling and clamping are monotone: pooling after them = them after pooling)
          // as v_med3_f32(x, y, +inf) -- a max without the operand quieting fmaxf pays under IEEE mode; one conversion.
          // (The float form below cost ~10 instructions per value; the e4m3 layers have HALF the items per tile of the bf16
          // ones at the same epilogue, so it weighed twice as much.)
          const float lo = a.relu ? 0.f : -448.f, inf = __builtin_inff();
          float q[2][16];
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              float s;
              if constexpr (M16) {  // (bias already inside the accumulators)
                s = acc4[i][r >> 3][u][(r >> 2) & 1][r & 3] * a.out_mul;
              } else {  // (nothing to clear: the next tile's first MFMAs start from C = 0)
                s = fmaf(acc[i][u][r], a.out_mul, bv[r] * a.out_mul);
              }
              q[i][r] = __builtin_amdgcn_fmed3f(s, lo, 448.f);
            }
          const int cofs8 = cur.co0 + cl;
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
            for (int i = 0; i < 2; ++i) {
              const u32x4 o = cvt16(q[i]);
              if (!EDGE || (py + i < H && px < W))
                *reinterpret_cast<u32x4*>(ymap + (((int64_t)cur.n * (H + 2) + (py + i + 1)) * Wp + (px + 1)) * Ob + cofs8) = o;
            }
          }
          if (a.pooled != nullptr) {
            float m[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float t = __builtin_amdgcn_fmed3f(q[0][r], q[1][r], inf);
              m[r] = __builtin_amdgcn_fmed3f(t, g4_dpp_swap1(t), inf);
            }
            const int Hq = H >> 1, Wq = W >> 1;
            const u32x4 o = cvt16(m);
            // every wave issues this instruction (even lanes store): the counted vmcnt at the next barrier relies on it
            if ((lane & 1) == 0 && (!EDGE || (py < H && px < W)))
              *reinterpret_cast<u32x4*>(qmap + (((int64_t)cur.n * (Hq + 2) + ((py >> 1) + 1)) * (Wq + 2) + ((px >> 1) + 1)) * Ob + cofs8) = o;
          }
          continue;
        }
        static_assert(OF8 || M16, "every form has its packed / e4m3 epilogue above");
      }
      // (an edge tile may skip store instructions -- a wave below the image stores nothing: no counted wait then)
      in_flight = (EDGE && (cur.y0 + C::TH > H || cur.x0 + C::TW > W)) ? 0 : nstores;
      bslot ^= 1;
    }
    G4_STAMP(3)  // tile epilogue done (stores issued)
#ifdef XV_CONV_TRACE
    ++trace_item;
#endif
    if (!has_next) break;
    ++items_done;
    lid = nlid;
    cur = nxt;
    chunk = nchunk;
    buf ^= 1;
  }
  XV_CLK_END(xv_clk_g4)
#ifdef XV_CONV_TRACE
  __syncthreads();
  if (!STATS && wave == 0 && (blockIdx.x & 31) == 0)
    for (int i = lane; i < 2 * 24 * 4; i += 64)
      xv_trace_buf4[(blockIdx.x >> 5) * (2 * 24 * 4) + i] = reinterpret_cast<long long*>(smem + C::LDS_BYTES_STAGE)[i];
#endif
  if constexpr (STATS) {
    // half-wave sums by DPP (row_shr 1, 2, 4, 8 inside each 16-lane row, row_bcast15 into the odd rows: lanes 31 / 63 hold
    // the totals), each wave's totals to its own LDS row, the 8 rows added in wave order (a fixed tree: reproducible bits),
    // then the workgroup's row of a.stats (zero outside its own output-channel tile)
#define G4_DPP_ADD(x, ctrl, rows) \
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), ctrl, rows, 0xf, true))
    float* st = reinterpret_cast<float*>(smem + C::STATS_OFF);
    const int cslot = cur.co0 >> 6;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float su = ssum[j][r], sq = ssq[j][r];
        G4_DPP_ADD(su, 0x111, 0xf);
        G4_DPP_ADD(sq, 0x111, 0xf);
        G4_DPP_ADD(su, 0x112, 0xf);
        G4_DPP_ADD(sq, 0x112, 0xf);
        G4_DPP_ADD(su, 0x114, 0xf);
        G4_DPP_ADD(sq, 0x114, 0xf);
        G4_DPP_ADD(su, 0x118, 0xf);
        G4_DPP_ADD(sq, 0x118, 0xf);
        if constexpr (M16) {  // lane 15 of each 16-lane row holds the total of the row's 16 channels-of-group-g sums
          if (j == 0 && n31 == 15) {
            st[wave * 128 + 16 * hh + r] = su;
            st[wave * 128 + 64 + 16 * hh + r] = sq;
          }
        } else {
          G4_DPP_ADD(su, 0x142, 0xa);
          G4_DPP_ADD(sq, 0x142, 0xa);
          if (n31 == 31) {
            st[wave * 128 + 32 * j + 16 * hh + r] = su;
            st[wave * 128 + 64 + 32 * j + 16 * hh + r] = sq;
          }
        }
      }
#undef G4_DPP_ADD
    __syncthreads();
    for (int i = tid; i < 2 * Cout; i += C::NT) {
      const int c = i < Cout ? i : i - Cout;
      float v = 0.f;
      if ((c >> 6) == cslot) {
        const int k = (i < Cout ? 0 : 64) + (c & 63);
#pragma unroll
        for (int w = 0; w < C::NWAVES; ++w) v += st[w * 128 + k];
      }
      a.stats[(int64_t)b * 2 * Cout + i] = v;
    }
  }
#undef G4_RD
#undef G4_LDW
#undef G4_LDPA
#undef G4_LDPB
#undef G4_WAIT_W
#undef G4_WAIT_WA
#undef G4_WAIT_WB
#undef G4_CAT
#undef G4_MFMA
#undef G4_MFMA_B
#undef G4_MFMA_Z
#undef G4_MFMA_C
#undef G4_C_ACC
#undef G4_C_BIAS
#undef G4_TAP_HEAD
#undef G4_TAP_HEAD0
#undef G4_TAP_BODY
#undef G4_DMA_PIECES
#undef G4_TAP
#undef G4_PIN
}

// generation-4 fp8 image: [tap][cin / 64][row rho over cout][64 B], 16-byte slots swizzled by g4_swz(rho & 63, slot), rows
// permuted inside every 32-row block (row 8g + 4h + q = channel 16h + 4g + q); one thread = 4 bytes
__global__ void pack_weights_f8_g4_kernel(const float* __restrict__ w, char* __restrict__ out, int taps, int cin, int cout,
                                          float mul) {
  const int64_t total4 = (int64_t)taps * cin * cout / 4;
  const int nch = cin >> 6;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total4; q += (int64_t)gridDim.x * blockDim.x) {
    const int64_t idx = q * 4;  // destination byte
    const int e = (int)(idx & 15);
    const int ps = (int)((idx >> 4) & 3);
    int64_t rest = idx >> 6;
    const int rho = (int)(rest % cout);
    rest /= cout;
    const int chunk = (int)(rest % nch);
    const int tap = (int)(rest / nch);
    const int m = rho & 31;
    const int co = (rho & ~31) + 16 * ((m >> 2) & 1) + 4 * (m >> 3) + (m & 3);
    const int ci = chunk * 64 + g4_swz(rho & 63, ps) * 16 + e;
    const float v0 = w[((int64_t)tap * cin + ci) * cout + co], v1 = w[((int64_t)tap * cin + ci + 1) * cout + co];
    const float v2 = w[((int64_t)tap * cin + ci + 2) * cout + co], v3 = w[((int64_t)tap * cin + ci + 3) * cout + co];
    *reinterpret_cast<uint32_t*>(out + idx) = g4_pack_fp8x4(v0, v1, v2, v3, mul);
  }
}

}  // namespace

// Can generation 4 run this shape?  (3x3, whole 64-byte chunks of input channels; any map size: partial tiles are
// handled by clamped DMA offsets and predicated stores)
bool xv_conv3x3_f8_dma_ok(int H, int W, int Cin, int Cout) {
  return H > 0 && W > 0 && Cin >= 64 && (Cin & 63) == 0 && (Cout & 63) == 0;
}
bool xv_conv3x3_dma4_bf16_ok(int H, int W, int Cin, int Cout) {
  return H > 0 && W > 0 && Cin >= 64 && (Cin & 31) == 0 && (Cout & 63) == 0;
}
// ... and does the map tile exactly in 16x32 pixels (no partial tiles)?
bool xv_conv3x3_dma4_exact(int H, int W) { return (H & 15) == 0 && (W & 31) == 0; }

namespace {
template <bool F8, bool OF8, bool STATS, bool EDGE, bool M16, bool DG = false, bool RT = false>
int g4_launch1(const F8Args& a, int grid, hipStream_t stream) {
  constexpr int lds = STATS ? G4::LDS_BYTES_STATS : G4::LDS_BYTES_STAGE + G4_TRACE_LDS;
  static_assert(lds <= 160 * 1024, "does not fit the LDS");
  static bool attr_set[XV_MAX_DEVICES] = {false};
  const hipError_t e =
      xv_allow_dynamic_lds(reinterpret_cast<const void*>(&conv_dma4_kernel<F8, OF8, STATS, EDGE, M16, DG, RT>), lds, attr_set);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL((conv_dma4_kernel<F8, OF8, STATS, EDGE, M16, DG, RT>), dim3((unsigned)grid), dim3(G4::NT), lds, stream, a);
  return xv_launch_status();
}
template <bool F8, bool OF8, bool STATS = false, bool M16 = false>
int g4_launch(const F8Args& a, int grid, hipStream_t stream) {
  if (STATS || xv_conv3x3_dma4_exact(a.H, a.W)) return g4_launch1<F8, OF8, STATS, false, M16>(a, grid, stream);
  return g4_launch1<F8, OF8, false, true, M16>(a, grid, stream);
}
}  // namespace

// in_f8 / out_f8: e4m3 input (map and weights) / output maps: (1, 1) = configuration 24; (0, 0) and (0, 1) = configuration
// 25, or 26 with m16 (the bf16 kernel on v_mfma_f32_16x16x32_bf16).  scale_x / out_mul as in ConvArgs.
int xv_launch_conv3x3_f8_dma(const void* x, const void* wpk, const float* bias, void* y, void* pooled, int N, int H, int W,
                             int Cin, int Cout, int relu, int in_f8, int out_f8, int scale_x, float out_mul, int num_cus,
                             hipStream_t stream, float* stats_rows, int m16, const void* mask, const void* addend) {
  if (!(in_f8 ? xv_conv3x3_f8_dma_ok(H, W, Cin, Cout) : xv_conv3x3_dma4_bf16_ok(H, W, Cin, Cout)) ||
      (y == nullptr && pooled == nullptr) || (in_f8 && !out_f8) || (in_f8 && m16) || (!in_f8 && !m16))
    return XV_ESHAPE;
  // the data-gradient epilogue exists in the packed epilogue of the bf16 16x16 form only
  if ((mask != nullptr || addend != nullptr) && (!m16 || out_f8 || stats_rows != nullptr || y == nullptr)) return XV_ESHAPE;
  F8Args a{};
  a.x = (const char*)x;
  a.wpk = (const char*)wpk;
  a.bias = bias;
  a.y = (char*)y;
  a.pooled = (char*)pooled;
  a.N = N, a.H = H, a.W = W, a.Cin = Cin, a.Cout = Cout;
  a.tiles_x = (W + G4::TW - 1) / G4::TW;
  a.tiles_y = (H + G4::TH - 1) / G4::TH;
  a.n_ct = Cout / 64;
  const int64_t ntiles = (int64_t)a.tiles_x * a.tiles_y * N * a.n_ct;
  if (ntiles <= 0 || ntiles > 0x7fffffff) return XV_ESHAPE;
  a.n_tiles = (int)ntiles;
  a.relu = relu;
  a.scale_x = scale_x;
  a.out_mul = out_mul;
  a.mask = (const char*)mask;
  a.addend = (const char*)addend;
  const int grid = num_cus > 0 ? num_cus : 256;
  if (stats_rows != nullptr) {  // per-channel sums of the outputs in the epilogue: bf16 maps, at most 8 channel tiles
    // exact tilings only (a partial tile's out-of-image pixels would enter the sums); one channel tile per workgroup
    if (in_f8 || out_f8 || Cout > 512 || !xv_conv3x3_dma4_exact(H, W) || (grid & 7) || (grid / 8) % a.n_ct) return XV_ESHAPE;
    a.stats = stats_rows;
    return g4_launch<false, false, true, true>(a, grid, stream);
  }
  if (mask != nullptr || addend != nullptr)  // the data-gradient kernel (16x16 form, bf16 map out)
    return xv_conv3x3_dma4_exact(H, W) ? g4_launch1<false, false, false, false, true, true>(a, grid, stream)
                                       : g4_launch1<false, false, false, true, true, true>(a, grid, stream);
  if (in_f8) return g4_launch<true, true>(a, grid, stream);
  return out_f8 ? g4_launch<false, true, false, true>(a, grid, stream) : g4_launch<false, false, false, true>(a, grid, stream);
}

// The routed pool of training (F8Args::route), bf16 maps that tile exactly in 16x32.  dgrad = 0: pooled map + route bytes
// [N][H/2][W/2][Cout] of relu(conv(x) + bias), no full map.  dgrad = 1: conv(x) (x = the gradient of the layer behind the
// pool, at the pooled size H x W) routed onto `out`, the bf16 map [N][2H+2][2W+2][Cout].
int xv_launch_conv3x3_dma4_route(const void* x, const void* wpk, const float* bias, void* out, void* route, int dgrad, int N,
                                 int H, int W, int Cin, int Cout, int num_cus, hipStream_t stream) {
  if (!xv_conv3x3_dma4_bf16_ok(H, W, Cin, Cout) || !xv_conv3x3_dma4_exact(H, W) || out == nullptr || route == nullptr)
    return XV_ESHAPE;
  F8Args a{};
  a.x = (const char*)x, a.wpk = (const char*)wpk, a.bias = bias;
  if (dgrad)
    a.y = (char*)out;
  else
    a.pooled = (char*)out;
  a.route = (char*)route;
  a.N = N, a.H = H, a.W = W, a.Cin = Cin, a.Cout = Cout;
  a.tiles_x = W / G4::TW;
  a.tiles_y = H / G4::TH;
  a.n_ct = Cout / 64;
  const int64_t ntiles = (int64_t)a.tiles_x * a.tiles_y * N * a.n_ct;
  if (ntiles <= 0 || ntiles > 0x7fffffff) return XV_ESHAPE;
  a.n_tiles = (int)ntiles;
  a.relu = dgrad ? 0 : 1;
  a.out_mul = 1.f;
  const int grid = num_cus > 0 ? num_cus : 256;
  return dgrad ? g4_launch1<false, false, false, false, true, true, true>(a, grid, stream)
               : g4_launch1<false, false, false, false, true, false, true>(a, grid, stream);
}

// Two problems of one shape in ONE launch (F8Args::n_first): the bf16 forward form on maps that tile exactly in 16x32.
// Both full maps or neither, both pooled maps or neither.
int xv_launch_conv3x3_dma4_pair(const void* const x[2], const void* const wpk[2], const float* const bias[2], void* const y[2],
                                void* const pooled[2], int N, int H, int W, int Cin, int Cout, int relu, int num_cus,
                                hipStream_t stream) {
  if (!xv_conv3x3_dma4_bf16_ok(H, W, Cin, Cout) || !xv_conv3x3_dma4_exact(H, W) || (y[0] == nullptr) != (y[1] == nullptr) ||
      (pooled[0] == nullptr) != (pooled[1] == nullptr) || (y[0] == nullptr && pooled[0] == nullptr))
    return XV_ESHAPE;
  F8Args a{};
  a.x = (const char*)x[0], a.wpk = (const char*)wpk[0], a.bias = bias[0], a.y = (char*)y[0], a.pooled = (char*)pooled[0];
  a.x2 = (const char*)x[1], a.wpk2 = (const char*)wpk[1], a.bias2 = bias[1], a.y2 = (char*)y[1], a.pooled2 = (char*)pooled[1];
  a.n_first = N;
  a.N = 2 * N, a.H = H, a.W = W, a.Cin = Cin, a.Cout = Cout;
  a.tiles_x = W / G4::TW;
  a.tiles_y = H / G4::TH;
  a.n_ct = Cout / 64;
  const int64_t ntiles = (int64_t)a.tiles_x * a.tiles_y * a.N * a.n_ct;
  if (ntiles <= 0 || ntiles > 0x7fffffff) return XV_ESHAPE;
  a.n_tiles = (int)ntiles;
  a.relu = relu;
  a.out_mul = 1.f;
  return g4_launch1<false, false, false, false, true>(a, num_cus > 0 ? num_cus : 256, stream);
}

// second image of the packed fp8 buffer (xv_pack_conv_weights_f8 calls this for 3x3 kernels with cin % 64 == 0)
void xv_launch_pack_weights_f8_g4(const float* w, char* out, int taps, int cin, int cout, float mul, hipStream_t stream) {
  const int64_t total4 = (int64_t)taps * cin * cout / 4;
  const int blocks = (int)((total4 + 255) / 256 < 4096 ? (total4 + 255) / 256 : 4096);
  hipLaunchKernelGGL(pack_weights_f8_g4_kernel, dim3(blocks), dim3(256), 0, stream, w, out, taps, cin, cout, mul);
}

#ifdef XV_CLOCK_STAMP
// [workgroup][s_memtime before, s_memrealtime before, s_memtime after, s_memrealtime after] of the last generation-4 launch
extern "C" int xv_debug_read_clock_g4(void* dst, size_t bytes) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(xv_clk_g4), bytes); }
extern "C" int xv_debug_reset_clock_g4(void) {
  static unsigned long long zeros[4 * XV_CLK_SLOTS];
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(xv_clk_g4), zeros, sizeof(zeros));
}
#endif

#ifdef XV_CONV_TRACE
extern "C" int xv_debug_read_trace4(void* dst, size_t bytes) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(xv_trace_buf4), bytes); }
#endif
