// conv2d forward (3x3 'same' and 1x1) as implicit GEMM on bf16 MFMA for gfx950.
//
// Replaces tf.layers.conv2d as called through xview/models/custom_layers.py:124-139 from
// xview/models/simple_fcn.py:39-79 (conv1_2 .. conv5_3, score_conv4, score_conv5).
//
// GEMM view:  D[cout][pixel] = sum_{tap, cin} Wt[cout][tap, cin] * X[pixel + tap][cin]
//   MFMA A operand = weights (rows = cout), B operand = activations (cols = pixels), so every
//   lane ends up holding 4 CONSECUTIVE output channels of one pixel (an 8-byte NHWC store).
// Tiling: a workgroup owns a (8*WR) x (16*WC) pixel patch of one image and 64*NW output channels;
//   each wave owns 8 rows x 16 pixels x 64 channels = 8 x 4 tiles of v_mfma_f32_16x16x32_bf16
//   (128 fp32 accumulators per lane).  Per 64-channel input chunk the (TH+2)x(TW+2) halo patch is
//   staged ONCE in LDS and re-read by all 9 taps; the 64*NW x 64 weight tile of each tap is
//   double-buffered.  Pixel rows / weight rows are 128 B in LDS with a 16-byte-slot XOR swizzle
//   (xv_swz) so the 16-lane ds_read_b128 groups are bank-conflict free.
#include <cstdlib>
#include <type_traits>

#include "xv_common.h"

#ifndef XV_CONV_PFD
#define XV_CONV_PFD 2
#endif
#ifndef XV_D2_BTAP0
#define XV_D2_BTAP0 1  // generation 2b: first tap that issues weight DMA pieces / pieces per tap (tuning)
#endif
#ifndef XV_D2_BPT
#define XV_D2_BPT 1
#endif
#ifndef XV_F8_BIG_MAP
#define XV_F8_BIG_MAP (1 << 30)  // pixels per launch from which the fp8 chooser takes configuration 16 (tools/conv_tune.py)
#endif

// conv1x1_gemm.hip
int xv_launch_conv1x1_gemm(const __bf16* x, const __bf16* wpk, const float* bias, __bf16* y, const __bf16* mask,
                           const __bf16* addend, int N, int H, int W, int Cin, int Cout, int relu, hipStream_t stream);
int xv_launch_conv1x1_n64(const __bf16* x, const __bf16* wpk, const float* bias, __bf16* y, const __bf16* mask,
                          const __bf16* addend, int N, int H, int W, int Cin, int Cout, int relu, hipStream_t stream);

// conv_f8_dma.hip (generation 4: the fp8 3x3 conv with all operands by LDS-DMA, configuration 24)
bool xv_conv3x3_f8_dma_ok(int H, int W, int Cin, int Cout);
bool xv_conv3x3_dma4_bf16_ok(int H, int W, int Cin, int Cout);
bool xv_conv3x3_dma4_exact(int H, int W);
int xv_launch_conv3x3_f8_dma(const void* x, const void* wpk, const float* bias, void* y, void* pooled, int N, int H, int W,
                             int Cin, int Cout, int relu, int in_f8, int out_f8, int scale_x, float out_mul, int num_cus,
                             hipStream_t stream, float* stats_rows = nullptr, int m16 = 0, const void* mask = nullptr,
                             const void* addend = nullptr);
int xv_launch_conv3x3_dma4_route(const void* x, const void* wpk, const float* bias, void* out, void* route, int dgrad, int N,
                                 int H, int W, int Cin, int Cout, int num_cus, hipStream_t stream);
void xv_launch_pack_weights_f8_g4(const float* w, char* out, int taps, int cin, int cout, float mul, hipStream_t stream);
// conv_col_dma.hip (generation 5: the generation-4 loop on a column of waves, 24x16 / 32x16 tiles, configurations 27 / 28)
bool xv_conv3x3_col_ok(int H, int W, int Cin, int Cout, int mt);
int xv_launch_conv3x3_col(const void* x, const void* wpk, const float* bias, void* y, void* pooled, int N, int H, int W, int Cin,
                          int Cout, int relu, int mt, int num_cus, hipStream_t stream, const void* mask = nullptr,
                          const void* addend = nullptr, void* split_ws = nullptr, size_t split_bytes = 0);
size_t xv_conv3x3_col_split_bytes(int N, int H, int W, int Cin, int Cout, int mt, int num_cus);
// two problems of one shape in one launch (generations 4 / 5)
int xv_launch_conv3x3_dma4_pair(const void* const x[2], const void* const wpk[2], const float* const bias[2], void* const y[2],
                                void* const pooled[2], int N, int H, int W, int Cin, int Cout, int relu, int num_cus,
                                hipStream_t stream);
int xv_launch_conv3x3_col_pair(const void* const x[2], const void* const wpk[2], const float* const bias[2], void* const y[2],
                               void* const pooled[2], int N, int H, int W, int Cin, int Cout, int relu, int mt, int num_cus,
                               hipStream_t stream);

namespace {

struct ConvArgs {
  const __bf16* x;
  const __bf16* wpk;
  const float* bias;
  __bf16* y;       // may be null
  __bf16* pooled;  // may be null
  const __bf16* mask;    // may be null: output zeroed where mask <= 0 (relu backward), same layout as y
  const __bf16* addend;  // may be null: added to the output before masking, same layout as y
  int N, H, W, Cin, Cout;
  int tiles_x, tiles_y, n_ct, n_tiles;
  int relu;
  int num_cus;
  // fp8 path (OCP e4m3fn): in_f8 = x and the packed weights are fp8 (kernel template F8); out_f8 = y / pooled are
  // written as fp8 of value * out_mul (= 2^-scale_exp of the output map) by the shared epilogue
  int in_f8, out_f8;
  int scale_x;    // E8M0 byte (127 + scale_exp of x) in all four bytes: the uniform block scale of the B operand
  float out_mul;
  int debug_same_patch;  // 0 except in -DXV_CONV_EXPERIMENTS tuning builds (configuration 21 only, see launch_conv_dma2)
  // generation 2, stream-K tail (see conv_dma_kernel): workspace of xv_conv2d_streamk_workspace_bytes() -- arrival counters
  // (zero between launches) + fp32 partial-tile slabs -- or null: every tile is computed whole by one workgroup
  char* sk_ws;
  // generation 5, split form (conv_col_dma.hip, MODE 2): fp32 slabs for the (tile, chunk group) items, or null
  void* split_ws;
  size_t split_bytes;
};

// stream-K workspace: 8 x 64 arrival counters (4 KB header), then two partial-tile slabs per workgroup of the grid
constexpr int XV_SK_HDR = 4096;
constexpr int XV_SK_SLAB = 8 * 16 * 1024;  // 8 waves x 16 accumulator tiles x (64 lanes x 16 B)
constexpr int XV_SK_MAX_CUS = 512;         // 64 workgroups per XCD group
constexpr int XV_SK_MAX_SPLIT = 4;

// Is the stream-K tail worth it for one XCD group's last round -- `rem` tiles of `nchunks` items on `nb` workgroups, slabs
// of `slab_kb` KB?  Returns the number of workgroups that share the tail's items (at most XV_SK_MAX_SPLIT per tile: the
// last arriver reads every other contributor's slab at 60-70 GB/s), or 0 for "compute every tile whole".  Measured on
// MI355X (bench.py --layer-profile, with / without XV_DMA_NO_STREAMK): an item takes ~2.9 us; the exchange costs ~8 us of
// latency chain (drain of the write-through stores, ticket, slab reads, one more epilogue) plus its bytes, written through
// and read back across XCDs, at ~2 TB/s chip-wide -- conv5_x at one image: 9 MB, 46 -> 27 us per launch; conv4_2 at one
// image: 32 MB, 47 -> 51 us; the half-round tails at 16 images: 24-32 MB for 8 items saved, no gain.  Hence: only where
// the items saved pay for the exchange 1.3 times over.
__host__ __device__ inline int xv_sk_parts(int rem, int nb, int nchunks, int slab_kb) {
  if (rem <= 0 || nchunks < 2) return 0;
  const int nbp = nb < rem * XV_SK_MAX_SPLIT ? nb : rem * XV_SK_MAX_SPLIT;
  const int saved = nchunks - (rem * nchunks + nbp - 1) / nbp;  // items off the critical path
  const int gain = saved * 290;                                  // us x 100
  const int cost = nbp * slab_kb * 100 / 125 + 800;              // 2 x (8 groups x nbp slabs) / 2 TB/s + 8 us, x 100
  return gain * 10 > cost * 13 ? nbp : 0;
}

typedef __attribute__((ext_vector_type(8))) int i32x8;

// four fp32 -> four e4m3 bytes (round-to-nearest-even), value * mul, saturating at the largest finite e4m3 (the
// conversion's own overflow behaviour depends on a mode bit, so the clamp is explicit)
__device__ __forceinline__ uint32_t pack_fp8x4(f32x4 v, float mul) {
  const float a = __builtin_amdgcn_fmed3f(v.x * mul, -448.f, 448.f);
  const float b = __builtin_amdgcn_fmed3f(v.y * mul, -448.f, 448.f);
  const float c = __builtin_amdgcn_fmed3f(v.z * mul, -448.f, 448.f);
  const float d = __builtin_amdgcn_fmed3f(v.w * mul, -448.f, 448.f);
  int p = 0;
  p = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, p, false);
  p = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, p, true);
  return (uint32_t)p;
}

// a lane's 4 consecutive output channels of one pixel: 8 bytes of bf16 or 4 bytes of e4m3 at ELEMENT offset eoff
__device__ __forceinline__ void store_out4(const ConvArgs& a, void* base, int64_t eoff, f32x4 v) {
  if (a.out_f8)
    *reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(base) + eoff) = pack_fp8x4(v, a.out_mul);
  else
    *reinterpret_cast<u32x2*>(reinterpret_cast<__bf16*>(base) + eoff) = u32x2{pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w)};
}


// ---- epilogue shared by both kernel generations: bias + relu, bf16, 8-byte NHWC stores (a lane holds 4
// consecutive output channels of pixel column px for MT rows), optional addend / relu mask (data gradient)
// and the fused 2x2 max-pool.  Zeroes the accumulators for the next tile.
__device__ __forceinline__ float dpp_swap1(float v) {
  // value of lane ^ 1 (quad_perm [1,0,3,2]) on the VALU, no LDS crossbar
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
}

template <int MT>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x4 (&acc)[MT][4], int n, int py0, int px, int cbase,
                                              int lane) {
  const int H = a.H, W = a.W, Cout = a.Cout, Wp = W + 2;
  f32x4 bj[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) bj[j] = *reinterpret_cast<const f32x4*>(a.bias + cbase + j * 16);
  if (a.y == nullptr && a.pooled != nullptr) {
    // pooled-only output: max first (bias add and relu are monotone, so they commute with it), then bias +
    // relu on a quarter of the values
    const int Hq = H >> 1, Wq = W >> 1;
    const int64_t qimg = (int64_t)n * (Hq + 2) * (Wq + 2) * Cout;
#pragma unroll
    for (int i = 0; i < MT; i += 2) {
      const int py = py0 + i;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x4 m;
        m.x = fmaxf(acc[i][j].x, acc[i + 1][j].x);
        m.y = fmaxf(acc[i][j].y, acc[i + 1][j].y);
        m.z = fmaxf(acc[i][j].z, acc[i + 1][j].z);
        m.w = fmaxf(acc[i][j].w, acc[i + 1][j].w);
        m.x = fmaxf(m.x, dpp_swap1(m.x));
        m.y = fmaxf(m.y, dpp_swap1(m.y));
        m.z = fmaxf(m.z, dpp_swap1(m.z));
        m.w = fmaxf(m.w, dpp_swap1(m.w));
        m += bj[j];
        if (a.relu) {
          m.x = fmaxf(m.x, 0.f);
          m.y = fmaxf(m.y, 0.f);
          m.z = fmaxf(m.z, 0.f);
          m.w = fmaxf(m.w, 0.f);
        }
        if ((lane & 1) == 0 && py < H && px < W)
          store_out4(a, a.pooled, qimg + ((int64_t)((py >> 1) + 1) * (Wq + 2) + ((px >> 1) + 1)) * Cout + cbase + j * 16, m);
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x4 v = acc[i][j] + bj[j];
        if (a.relu) {
          v.x = fmaxf(v.x, 0.f);
          v.y = fmaxf(v.y, 0.f);
          v.z = fmaxf(v.z, 0.f);
          v.w = fmaxf(v.w, 0.f);
        }
        acc[i][j] = v;
      }
    if (a.y != nullptr) {
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int py = py0 + i;
        if (py < H && px < W) {
          const int64_t off = (int64_t)n * (H + 2) * Wp * Cout + ((int64_t)(py + 1) * Wp + (px + 1)) * Cout + cbase;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            f32x4 v = acc[i][j];
            if (a.addend != nullptr) {
              const u32x2 ad = *reinterpret_cast<const u32x2*>(a.addend + off + j * 16);
              v.x += bf16_bits_to_f32(ad.x & 0xffffu);
              v.y += __builtin_bit_cast(float, ad.x & 0xffff0000u);
              v.z += bf16_bits_to_f32(ad.y & 0xffffu);
              v.w += __builtin_bit_cast(float, ad.y & 0xffff0000u);
            }
            if (a.mask != nullptr) {
              const u32x2 mk = *reinterpret_cast<const u32x2*>(a.mask + off + j * 16);
              v.x = bf16_bits_to_f32(mk.x & 0xffffu) > 0.f ? v.x : 0.f;
              v.y = __builtin_bit_cast(float, mk.x & 0xffff0000u) > 0.f ? v.y : 0.f;
              v.z = bf16_bits_to_f32(mk.y & 0xffffu) > 0.f ? v.z : 0.f;
              v.w = __builtin_bit_cast(float, mk.y & 0xffff0000u) > 0.f ? v.w : 0.f;
            }
            store_out4(a, a.y, off + j * 16, v);
          }
        }
      }
    }
    if (a.pooled != nullptr) {
      // fused max_pooling2d(2,2): rows (i, i+1) live in this lane, columns (px, px^1) in lanes l, l^1
      const int Hq = H >> 1, Wq = W >> 1;
      const int64_t qimg = (int64_t)n * (Hq + 2) * (Wq + 2) * Cout;
#pragma unroll
      for (int i = 0; i < MT; i += 2) {
        const int py = py0 + i;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          f32x4 m;
          m.x = fmaxf(acc[i][j].x, acc[i + 1][j].x);
          m.y = fmaxf(acc[i][j].y, acc[i + 1][j].y);
          m.z = fmaxf(acc[i][j].z, acc[i + 1][j].z);
          m.w = fmaxf(acc[i][j].w, acc[i + 1][j].w);
          m.x = fmaxf(m.x, dpp_swap1(m.x));
          m.y = fmaxf(m.y, dpp_swap1(m.y));
          m.z = fmaxf(m.z, dpp_swap1(m.z));
          m.w = fmaxf(m.w, dpp_swap1(m.w));
          if ((lane & 1) == 0 && py < H && px < W)
            store_out4(a, a.pooled, qimg + ((int64_t)((py >> 1) + 1) * (Wq + 2) + ((px >> 1) + 1)) * Cout + cbase + j * 16, m);
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
}

template <int MT, int WR, int WC, int NW, int KS, int TPS_ = 1>
struct ConvCfg {
  static constexpr int TPS = (KS == 3) ? TPS_ : 1;  // taps per barrier stage
  static constexpr int NT = 64 * WR * WC * NW;
  static constexpr int TH = MT * WR, TW = 16 * WC;
  static constexpr int HALO = (KS == 3) ? 1 : 0;
  static constexpr int HH = TH + 2 * HALO, HW = TW + 2 * HALO;
  static constexpr int NPIX = HH * HW;
  static constexpr int BN = 64 * NW;
  static constexpr int A_BYTES = ((NPIX * 128 + 255) / 256) * 256;
  static constexpr int B_BYTES = BN * 128;
  static constexpr int NTAPS = KS * KS;
  static constexpr int NST = (NTAPS + TPS - 1) / TPS;  // stages per input chunk
  static constexpr int LDS_BYTES = A_BYTES + 2 * TPS * B_BYTES;
  static constexpr int A_ITERS = (NPIX * 8 + NT - 1) / NT;
  static constexpr int B_ITERS = B_BYTES / 16 / NT;
  static_assert(B_BYTES % (16 * NT) == 0, "weight tile must split evenly over the threads");
};

// OCC = waves per SIMD the register allocation is bounded for (2 -> <= 256 VGPRs, two 4-wave
// workgroups per CU; 1 -> up to 512).  The next-item patch prefetch (PFA) keeps A_ITERS*4 extra
// registers live under the last tap, which only fits with MT = 4 or OCC = 1.
//
// Persistent workgroups: the grid is min(#tiles, resident slots); a workgroup walks its share of
// the (pixel patch, cout tile) list of its XCD, and while the last tap of one work item (= one
// 64-channel input chunk of one tile) runs on the matrix cores the next item's activation patch and
// first weight tile are already in flight to registers -- across chunk AND tile boundaries.
// All LDS fragment reads use a per-lane base register + compile-time immediate offset (taps are
// fully unrolled), so the inner loop issues no address arithmetic.
// DMAB: the weight tiles of a stage stream global -> LDS by LDS-DMA (`global_load_lds_dwordx4`, a linear copy
// because the packed image is pre-swizzled) instead of global -> VGPR -> ds_write: no staging registers and,
// more importantly, 63 % fewer bytes through the VGPR->LDS store path, which together with the fragment
// reads keeps the LDS ~80 % busy in the register-staged form.  The DMA of stage s+1 is issued at the start
// of stage s BEFORE any patch prefetch loads, so a counted `s_waitcnt vmcnt(#patch loads)` at the end of the
// stage retires exactly the DMA (VMEM operations complete in order) and leaves the patch in flight across
// the raw `s_barrier`.  Stage buffers alternate with a per-item parity because a chunk has an odd number of
// stages.
// F8: the fp8 form (BASELINE config "fp8 MFMA conv path").  A 128-channel chunk of e4m3 bytes has the byte geometry of
// a 64-channel bf16 chunk (128-byte pixel / weight rows), so staging, LDS images, swizzle and stage pipeline are shared;
// a lane's fragment is the SAME two 16-byte slots it reads for the two bf16 k-halves, concatenated into the 32 bytes
// v_mfma_scale_f32_16x16x128_f8f6f4 wants: lane (row r, group g) holds channels [16g, 16g+16) and [64+16g, 64+16g+16)
// of the chunk for the A (weights) and the B (pixels) operand alike, so every product pairs equal channels whatever
// k index the hardware gives a byte.  One MFMA per (row, channel block) and tap instead of two, at twice the K and
// twice the cycles of the bf16 instruction = 2x the FLOP per clock (MI355X_MICROARCH.md, Matrix cores).  The per-tensor
// power-of-two scales of the two operands ride in as uniform E8M0 block scales: accumulators are in real units.
template <int MT, int WR, int WC, int NW, int KS, int OCC, int TPS, int DMAB, int F8 = 0>
__global__ __launch_bounds__(64 * WR * WC * NW, OCC) void conv_mfma_kernel(ConvArgs a) {
  using C = ConvCfg<MT, WR, WC, NW, KS, TPS>;
  constexpr int ESZ = F8 ? 1 : 2;  // bytes per activation element
  // (fp8: fragments are twice as wide and the cross-item prefetch no longer fits 256 registers -- it spilled 500+)
  // F8 == 2: the fp8 kernel WITH the prefetch (tuning variant, XV_F8_PFA=1)
  constexpr bool PFA = (F8 != 1) && ((MT == 4) || (OCC == 1));
  constexpr int PFD = XV_CONV_PFD;  // how many stages before the end of an item its successor's patch is requested
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const As = smem;
  char* const Bs = smem + C::A_BYTES;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave % NW;
  const int wm = wave / NW;
  const int wr = wm / WC;
  const int wc = wm % WC;
  const int l15 = lane & 15, lg = lane >> 4;

  const int H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout;
  const int Wp = W + 2;
  const int nchunks = F8 ? Cin >> 7 : Cin >> 6;  // 128 bytes of input channels per chunk
  // fp8 weights: [256-byte header: int32 scale exponent][image]
  const char* const wimg = reinterpret_cast<const char*>(a.wpk) + (F8 ? 256 : 0);
  int scale_w = 0;
  if constexpr (F8) scale_w = ((127 + *reinterpret_cast<const int*>(a.wpk)) & 0xff) * 0x01010101;
  const bool f8_prio = !(a.debug_same_patch & 32);  // (bit 5: XV_F8_NO_PRIO, A/B timing)

  // ---- this workgroup's share of the tile list (XCD-aware, placement affects speed only) --------
  // workgroups b, b+8, ... share an XCD (round-robin dispatch); each XCD owns a contiguous range of
  // logical tile ids, in which consecutive ids are the cout tiles of ONE pixel patch (shared L2 lines).
  const int G = gridDim.x, b = blockIdx.x;
  const int xcd = b & 7, bi = b >> 3;
  const int nb = (G - xcd + 7) >> 3;  // workgroups of this XCD
  const int T = a.n_tiles;
  const int tq = T >> 3, trm = T & 7;
  const int t_begin = xcd * tq + (xcd < trm ? xcd : trm);
  const int t_end = t_begin + tq + (xcd < trm ? 1 : 0);

  struct Tile {
    int n, y0, x0, co0;
  };
  auto decode = [&](int lid) {
    Tile t;
    t.co0 = (lid % a.n_ct) * C::BN;
    int r = lid / a.n_ct;
    t.x0 = (r % a.tiles_x) * C::TW;
    r /= a.tiles_x;
    t.y0 = (r % a.tiles_y) * C::TH;
    t.n = r / a.tiles_y;
    return t;
  };

  // ---- per-lane LDS fragment bases (16-byte slot s of pixel column hx / weight row n lives at
  //      slot s ^ ((hx or n) & 6): conflict-free ds_read_b128 for 16 consecutive columns) ----
  int abase[KS][2];
#pragma unroll
  for (int dx = 0; dx < KS; ++dx)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int hx = wc * 16 + l15 + dx;
      abase[dx][kk] = ((wr * MT) * C::HW + hx) * 128 + (xv_swz(hx, kk * 4 + lg) << 4);
    }
  int wbase[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) wbase[kk] = C::A_BYTES + (wn * 64 + l15) * 128 + (xv_swz(l15, kk * 4 + lg) << 4);

  // ---- staging: global -> registers (issued early), registers -> LDS (after the barrier) ---------
  // Patch coordinates are clamped onto the zero border of the padded buffer: every load is in-bounds
  // and unconditional; outputs fed by clamped pixels are never stored.
  auto a_load = [&](const Tile& t, int chunk, auto& v, auto IT0, auto IT1) {
    const char* ximg = reinterpret_cast<const char*>(a.x) + (int64_t)t.n * (H + 2) * Wp * Cin * ESZ + chunk * 128;
#pragma unroll
    for (int it = IT0; it < IT1; ++it) {
      int idx = tid + it * C::NT;
      idx = idx < C::NPIX * 8 ? idx : C::NPIX * 8 - 1;
      const int p = idx >> 3, s = idx & 7;
      const int hy = p / C::HW, hx = p - hy * C::HW;
      int yy = t.y0 + hy + (1 - C::HALO), xx = t.x0 + hx + (1 - C::HALO);  // padded coords
      yy = yy < H + 1 ? yy : H + 1;
      xx = xx < W + 1 ? xx : W + 1;
      v[it - IT0] = *reinterpret_cast<const u32x4*>(ximg + ((int64_t)yy * Wp + xx) * Cin * ESZ + s * 16);
    }
  };
  auto a_store = [&](const auto& v, auto IT0, auto IT1) {
#pragma unroll
    for (int it = IT0; it < IT1; ++it) {
      const int idx = tid + it * C::NT;
      const int p = idx >> 3, s = idx & 7;
      const int hx = p % C::HW;
      if (idx < C::NPIX * 8) *reinterpret_cast<u32x4*>(As + p * 128 + (xv_swz(hx, s) << 4)) = v[it - IT0];
    }
  };
  // a stage = C::TPS consecutive taps sharing one barrier; its weight tiles sit back to back in LDS
  auto b_load = [&](int co0, int stage, int chunk, u32x4(&v)[C::TPS * C::B_ITERS]) {
#pragma unroll
    for (int tt = 0; tt < C::TPS; ++tt) {
      const int tap = stage * C::TPS + tt;
      if (tap < C::NTAPS) {
        const char* src = wimg + (((int64_t)(tap * nchunks + chunk) * Cout + co0) << 7);
#pragma unroll
        for (int it = 0; it < C::B_ITERS; ++it)
          v[tt * C::B_ITERS + it] = *reinterpret_cast<const u32x4*>(src + ((tid + it * C::NT) << 4));
      }
    }
  };
  auto b_store = [&](int buf, int stage, const u32x4(&v)[C::TPS * C::B_ITERS]) {
    char* dst = Bs + buf * (C::TPS * C::B_BYTES);
#pragma unroll
    for (int tt = 0; tt < C::TPS; ++tt)
      if (stage * C::TPS + tt < C::NTAPS) {
#pragma unroll
        for (int it = 0; it < C::B_ITERS; ++it)
          *reinterpret_cast<u32x4*>(dst + tt * C::B_BYTES + ((tid + it * C::NT) << 4)) = v[tt * C::B_ITERS + it];
      }
  };

  // LDS-DMA of one stage's weight tiles (TPS taps x B_BYTES, each a contiguous run of the packed image)
  constexpr int STAGE_BYTES = C::TPS * C::B_BYTES;
  constexpr int DMA_PER_TAP = C::B_BYTES / 1024;  // 1 KB per wave-instruction
  constexpr int NWAVES = C::NT / 64;
  auto b_dma = [&](int co0, int stage, int chunk, int buf) {
#pragma unroll
    for (int tt = 0; tt < C::TPS; ++tt) {
      const int tap = stage * C::TPS + tt;
      if (tap < C::NTAPS) {
        // in assembly (SGPR base + lane offset): the builtin makes hipcc model a FLAT access, after which every
        // LDS wait it inserts is lgkmcnt(0) instead of a counted one
        const char* src = wimg + (((int64_t)(tap * nchunks + chunk) * Cout + co0) << 7);
        const int dst = C::A_BYTES + buf * STAGE_BYTES + tt * C::B_BYTES;
#pragma unroll
        for (int i = 0; i < (DMA_PER_TAP + NWAVES - 1) / NWAVES; ++i) {
          const int piece = wave + i * NWAVES;
          if (piece < DMA_PER_TAP)
            asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst + piece * 1024), "v"(lane * 16),
                         "s"(src + piece * 1024)
                         : "memory");
        }
      }
    }
  };

  int lid = t_begin + bi;
  if (lid >= t_end) return;
  Tile cur = decode(lid);
  int chunk = 0;

  f32x4 acc[MT][4];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  u32x4 areg[PFA ? C::A_ITERS : 1];
  u32x4 breg[C::TPS * C::B_ITERS];
  constexpr std::integral_constant<int, 0> I0{};
  constexpr std::integral_constant<int, (C::A_ITERS + 1) / 2> IH{};
  constexpr std::integral_constant<int, C::A_ITERS> IN{};
  int par = 0;  // DMAB: stage s of the current item lives in weight buffer (s + par) & 1
  if constexpr (DMAB) b_dma(cur.co0, 0, 0, 0);
  if constexpr (PFA) a_load(cur, 0, areg, I0, IN);
  if constexpr (!DMAB) b_load(cur.co0, 0, 0, breg);
  bool first = true;

  while (true) {
    if (!first) __syncthreads();  // every wave is done reading As / Bs of the previous work item
    first = false;
    if constexpr (PFA) {
      a_store(areg, I0, IN);
    } else {
      // no registers to spare for a prefetch: stage the patch in two halves
      u32x4 atmp[(C::A_ITERS + 1) / 2];
      a_load(cur, chunk, atmp, I0, IH);
      a_store(atmp, I0, IH);
      a_load(cur, chunk, atmp, IH, IN);
      a_store(atmp, IH, IN);
    }
    if constexpr (!DMAB) b_store(0, 0, breg);
    // DMAB: this item's first weight stage was requested at the start of the previous item's last stage, AFTER that
    // item's patch prefetch -- the compiler's counted waits for the patch registers (it cannot see the asm DMA) do
    // not cover it, and __syncthreads() waits for LDS operations only (s_waitcnt lgkmcnt(0); s_barrier).  Without this
    // explicit wait a wave could pass the barrier with its DMA pieces still in flight: seen as wrong outputs once in
    // ~100 launches on freshly mapped weights (cold TLB: a slow DMA) -- every other launch read the stage in time.
    if constexpr (DMAB) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // the work item after this one: next chunk of this tile, else chunk 0 of this workgroup's next tile
    const bool last_chunk = chunk + 1 == nchunks;
    const int nlid = last_chunk ? lid + nb : lid;
    const bool has_next = nlid < t_end;
    const Tile nxt = (last_chunk && has_next) ? decode(nlid) : cur;
    const int nchunk = last_chunk ? 0 : chunk + 1;

#pragma unroll
    for (int st = 0; st < C::NST; ++st) {
      const int cb = DMAB ? ((st + par) & 1) * STAGE_BYTES : (st & 1) * STAGE_BYTES;
      constexpr int PF_STAGE = C::NST - 1 - PFD > 0 ? C::NST - 1 - PFD : 0;
      if constexpr (DMAB) {
        // next stage's weights first (oldest VMEM operation of this stage) ...
        if (st + 1 < C::NST)
          b_dma(cur.co0, st + 1, chunk, (st + 1 + par) & 1);
        else if (has_next)
          b_dma(nxt.co0, 0, nchunk, (st + 1 + par) & 1);
        // the counted wait below relies on the DMA being issued BEFORE the patch loads: pin the order
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
      }
      // the next work item's patch is requested PFD stages before the end of this one (its HBM/L2 latency
      // must be covered by MFMA work: one stage is only ~1-2k cycles), its first weight tile in the last stage
      if constexpr (PFA) {
        if (st == PF_STAGE && has_next) a_load(nxt, nchunk, areg, I0, IN);
      }
      if constexpr (!DMAB) {
        if (st + 1 < C::NST) {
          b_load(cur.co0, st + 1, chunk, breg);
        } else if (has_next) {
          b_load(nxt.co0, 0, nchunk, breg);
        }
      }
#pragma unroll
      for (int tt = 0; tt < C::TPS; ++tt) {
        const int tap = st * C::TPS + tt;
        if (tap < C::NTAPS) {
          const int dy = (KS == 3) ? tap / 3 : 0;
          const int dx = (KS == 3) ? tap % 3 : 0;
          if constexpr (F8) {
            // Fragment reads in assembly, waited for by hand: left to the compiler, the 32-byte fragments of several
            // taps were hoisted above the MFMAs and 400 registers (accumulators included) went to scratch.  One tap =
            // 16 ds_read_b128 (W: 4 channel blocks x 2 slots, X: MT rows x 2 slots), one wait, 4*MT MFMAs of 32 cycles;
            // the SIMD's other wave (two 4-wave workgroups per CU, or the 8-wave tile) computes during the reads.
            // Addresses are LDS byte offsets (the dynamic segment starts at 0: the kernel has no static LDS).
            u32x4 wlo[4], whi[4], xlo[MT], xhi[MT];
            const int wa0 = wbase[0] + cb, wa1 = wbase[1] + cb;
#define XV_F8_RD(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              XV_F8_RD(wlo[j], wa0, tt * C::B_BYTES + j * 2048);
              XV_F8_RD(whi[j], wa1, tt * C::B_BYTES + j * 2048);
            }
#pragma unroll
            for (int i = 0; i < MT; ++i) {
              XV_F8_RD(xlo[i], abase[dx][0], (i + dy) * (C::HW * 128));
              XV_F8_RD(xhi[i], abase[dx][1], (i + dy) * (C::HW * 128));
            }
#undef XV_F8_RD
            static_assert(MT == 4, "operand list of the wait below");
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(wlo[0]), "+v"(whi[0]), "+v"(wlo[1]), "+v"(whi[1]), "+v"(wlo[2]), "+v"(whi[2]), "+v"(wlo[3]),
                           "+v"(whi[3]), "+v"(xlo[0]), "+v"(xhi[0]), "+v"(xlo[1]), "+v"(xhi[1]), "+v"(xlo[2]), "+v"(xhi[2]),
                           "+v"(xlo[3]), "+v"(xhi[3]));
            __builtin_amdgcn_sched_barrier(0);
            auto cat = [](const u32x4 lo, const u32x4 hi) {
              return i32x8{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
            };
            // burst priority (the generation-2 / filter-gradient scheme): the tap's 4 MT MFMAs run as one burst at raised
            // priority, so the SIMD's two waves alternate tap by tap -- one bursts while the other's fragment reads are in
            // flight -- instead of interleaving MFMA by MFMA
#pragma unroll
            for (int i = 0; i < MT; ++i) {
              const i32x8 xf = cat(xlo[i], xhi[i]);
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(cat(wlo[j], whi[j]), xf, acc[i][j], 0, 0, 0,
                                                                            scale_w, 0, a.scale_x);
                if (i == 0 && j == 0 && f8_prio) __builtin_amdgcn_s_setprio(2);
              }
            }
            if (f8_prio) __builtin_amdgcn_s_setprio(0);
            // An MFMA is a pure value to the instruction selector: nothing orders it against the (chained) asm reads of
            // the next tap, and it sank below them -- every tap's fragments then lived until the end of the item.  An
            // empty asm that "modifies" each accumulator pins this tap's MFMAs in front of the next tap's reads.
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
              for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(acc[i][j]));
            __builtin_amdgcn_sched_barrier(0);
          } else {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
              bf16x8 wf[4];
#pragma unroll
              for (int j = 0; j < 4; ++j)
                wf[j] = *reinterpret_cast<const bf16x8*>(smem + wbase[kk] + cb + tt * C::B_BYTES + j * 2048);
#pragma unroll
              for (int i = 0; i < MT; ++i) {
                const bf16x8 xf = *reinterpret_cast<const bf16x8*>(smem + abase[dx][kk] + (i + dy) * (C::HW * 128));
#pragma unroll
                for (int j = 0; j < 4; ++j)
                  acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf, acc[i][j], 0, 0, 0);
              }
            }
          }
        }
      }
      if (st + 1 < C::NST) {
        if constexpr (DMAB) {
          // retire this stage's DMA but not the patch loads issued after it (in-order VMEM completion);
          // in every other stage nothing newer than the DMA is outstanding -> vmcnt(0)
          if (PFA && st == PF_STAGE && has_next) {
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(C::A_ITERS) : "memory");
          } else {
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
          }
        } else {
          b_store((st & 1) ^ 1, st + 1, breg);
          __syncthreads();
        }
      }
    }
    if constexpr (DMAB) par = (par + C::NST) & 1;

    if (last_chunk) {
      conv_epilogue<MT>(a, acc, cur.n, cur.y0 + wr * MT, cur.x0 + wc * 16 + l15, cur.co0 + wn * 64 + lg * 4, lane);
    }
    if (!has_next) break;
    lid = nlid;
    cur = nxt;
    chunk = nchunk;
  }
}

template <int MT, int WR, int WC, int NW, int KS, int OCC, int TPS = 1, int DMAB = 0, int F8 = 0>
int launch_conv(const ConvArgs& a0, hipStream_t stream) {
  using C = ConvCfg<MT, WR, WC, NW, KS, TPS>;
  ConvArgs a = a0;
  static const bool f8_no_prio = getenv("XV_F8_NO_PRIO") != nullptr;
  if (F8 && f8_no_prio) a.debug_same_patch |= 32;
  a.tiles_x = (a.W + C::TW - 1) / C::TW;
  a.tiles_y = (a.H + C::TH - 1) / C::TH;
  a.n_ct = a.Cout / C::BN;
  static bool attr_set[XV_MAX_DEVICES] = {false};
  {
    const hipError_t e = xv_allow_dynamic_lds(reinterpret_cast<const void*>(&conv_mfma_kernel<MT, WR, WC, NW, KS, OCC, TPS, DMAB, F8>),
                                              C::LDS_BYTES, attr_set);
    if (e != hipSuccess) return (int)e;
  }
  const int64_t ntiles = (int64_t)a.tiles_x * a.tiles_y * a.N * a.n_ct;
  if (ntiles <= 0 || ntiles > 0x7fffffff) return XV_ESHAPE;
  if (a.pooled && (C::TH & 1)) return XV_ESHAPE;
  a.n_tiles = (int)ntiles;
  // resident workgroups: limited by LDS (160 KB / CU) and by the register bound OCC
  constexpr int by_lds = (160 * 1024) / C::LDS_BYTES;
  constexpr int by_reg = (OCC * 4 * 64) / C::NT > 0 ? (OCC * 4 * 64) / C::NT : 1;
  constexpr int per_cu = by_lds < by_reg ? by_lds : by_reg;
  const int64_t slots = (int64_t)a.num_cus * per_cu;
  const int64_t nblk = ntiles < slots ? ntiles : slots;
  hipLaunchKernelGGL((conv_mfma_kernel<MT, WR, WC, NW, KS, OCC, TPS, DMAB, F8>), dim3((unsigned)nblk), dim3(C::NT), C::LDS_BYTES, stream, a);
  return xv_launch_status();
}


// =================================================================================================
// Generation 2: every operand reaches LDS by LDS-DMA; 32-channel input chunks.
//
// One 8-wave workgroup per CU owns a (4*WR) x (16*WC) pixel patch x 64 output channels; a work item is one
// 32-channel chunk of one tile = 9 taps x 16 MFMAs per wave (K = 32 is exactly one v_mfma_f32_16x16x32_bf16).
// With 64-byte pixel rows the halo patch of a 16x32 tile is 38 KB and ALL nine 64x32 weight tiles of the
// item are 36 KB, so both are double-buffered whole (150 KB LDS): the DMA of item i+1 is issued right after
// the single barrier that starts item i and has the full item (~2.3 k MFMA cycles per wave) to land.  No
// VGPR staging, no ds_write, one barrier per item; fragment reads are software-pipelined one tap ahead in
// a second register set.
// LDS images: pixel p = hy*HW + hx at p*64, weight row n at n*64 (per tap 4 KB); 16-byte slot s lives at
// s ^ ((hx or n) >> 1 & 2): in every 16-lane ds_read_b128 group the four lanes that share an address
// residue mod 4 rows sit 4 rows apart with logical slots {a, a^1, a^1, a} -> physical {a, a^3, a^1, a^2}:
// conflict-free for any start column, and independent of the patch row (immediates for dy / row offsets).
//
// Epilogue in two phases.  Loads, stores and LDS-DMA share one in-order counter (vmcnt) and one in-order queue
// per CU, so a tile stored in one burst at its end (a) is waited for together with the next item's operands
// and (b) holds up those operands' DMA behind 128 store instructions -- measured 12-70 % of the kernel on the
// layers that write full-resolution maps.  Instead the tile is only CONVERTED at its end (bias from LDS,
// relu, addend, mask, pool, bf16 pack; v_permlane16_swap pairs the 4-channel groups of lanes l and l+16 into
// 16-byte pieces -> half as many stores, 64 contiguous bytes per pixel), and the 8 (pooled: 4) store
// instructions are issued one per tap during the NEXT work item, behind that item's DMA; the wait that
// ends the item is a counted vmcnt(#stores), which retires the DMA and leaves the stores in flight.
struct DmaPend {
  int n, y0, x0, co0;
  bool on;
};


__device__ __forceinline__ uint32_t dpp_swap1_u32(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xf, 0xf, true);
}

// Conversion phase: bias (+ addend, mask) in fp32 and rounding to bf16 pairs -- all eight waves of the workgroup
// do this at the same moment, so nothing hides it and it is kept minimal: relu and the lane pairing of the map
// that will be stored are left to the store phase, where they ride between the MFMAs of the next work item.
// pq[i][j] = the two bf16 pairs of row i, channel block j (pooled-only: pooled row i of rows 2i, 2i+1).
// Does NOT clear the accumulators: the first tap of the next tile starts from C = 0.
template <int MT>
__device__ __forceinline__ void dma_epilogue_pack(const ConvArgs& a, f32x4 (&acc)[MT][4], u32x2 (&pq)[MT][4],
                                                  const float* bias_lds, int n, int py0, int px, int cbase, int lane) {
  const int H = a.H, W = a.W, Cout = a.Cout, Wp = W + 2;
  const int lg = lane >> 4;
  f32x4 bj[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) bj[j] = *reinterpret_cast<const f32x4*>(bias_lds + lg * 4 + j * 16);
  // data-gradient extras: all addend / mask words of the tile are requested before the first is used (one
  // memory round trip instead of 32 dependent ones)
  u32x2 ad[MT][4], mk[MT][4];
  if (a.addend != nullptr) {
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const bool in = py0 + i < H && px < W;  // edge tiles: pixels past the image read the image's first pixel instead
      const int64_t off = (int64_t)n * (H + 2) * Wp * Cout + ((int64_t)((in ? py0 + i : 0) + 1) * Wp + ((in ? px : 0) + 1)) * Cout + cbase;
#pragma unroll
      for (int j = 0; j < 4; ++j) ad[i][j] = *reinterpret_cast<const u32x2*>(a.addend + off + j * 16);
    }
  }
  if (a.mask != nullptr) {
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const bool in = py0 + i < H && px < W;
      const int64_t off = (int64_t)n * (H + 2) * Wp * Cout + ((int64_t)((in ? py0 + i : 0) + 1) * Wp + ((in ? px : 0) + 1)) * Cout + cbase;
#pragma unroll
      for (int j = 0; j < 4; ++j) mk[i][j] = *reinterpret_cast<const u32x2*>(a.mask + off + j * 16);
    }
  }
  u32x2 h[MT][4];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x4 v = acc[i][j] + bj[j];
      if (a.addend != nullptr) {
        v.x += bf16_bits_to_f32(ad[i][j].x & 0xffffu);
        v.y += __builtin_bit_cast(float, ad[i][j].x & 0xffff0000u);
        v.z += bf16_bits_to_f32(ad[i][j].y & 0xffffu);
        v.w += __builtin_bit_cast(float, ad[i][j].y & 0xffff0000u);
      }
      if (a.mask != nullptr) {
        v.x = bf16_bits_to_f32(mk[i][j].x & 0xffffu) > 0.f ? v.x : 0.f;
        v.y = __builtin_bit_cast(float, mk[i][j].x & 0xffff0000u) > 0.f ? v.y : 0.f;
        v.z = bf16_bits_to_f32(mk[i][j].y & 0xffffu) > 0.f ? v.z : 0.f;
        v.w = __builtin_bit_cast(float, mk[i][j].y & 0xffff0000u) > 0.f ? v.w : 0.f;
      }
      h[i][j] = u32x2{pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w)};
    }
  if (a.pooled == nullptr) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) pq[i][j] = h[i][j];
    return;
  }
  // fused max_pooling2d(2,2): rows (i, i+1) live in this lane, columns (px, px^1) in lanes l, l^1 (even MT only: the
  // launcher refuses a pooled output for the 3-row configuration)
  if constexpr (MT % 2 != 0) {
    return;
  } else {
  u32x2 m[MT / 2][4];
  if (a.relu) {
    // relu first (once more at store time: idempotent), then the integer max is the float max
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) h[i][j] = u32x2{pk_max_i16(h[i][j].x, 0u), pk_max_i16(h[i][j].y, 0u)};
#pragma unroll
    for (int i = 0; i < MT; i += 2)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        u32x2 t = u32x2{pk_max_i16(h[i][j].x, h[i + 1][j].x), pk_max_i16(h[i][j].y, h[i + 1][j].y)};
        m[i >> 1][j] = u32x2{pk_max_i16(t.x, dpp_swap1_u32(t.x)), pk_max_i16(t.y, dpp_swap1_u32(t.y))};
      }
  } else {
    // signed inputs: the integer order is wrong for negative values, pool in fp32 (bias add is monotone)
#pragma unroll
    for (int i = 0; i < MT; i += 2)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x4 t;
        t.x = fmaxf(acc[i][j].x, acc[i + 1][j].x);
        t.y = fmaxf(acc[i][j].y, acc[i + 1][j].y);
        t.z = fmaxf(acc[i][j].z, acc[i + 1][j].z);
        t.w = fmaxf(acc[i][j].w, acc[i + 1][j].w);
        t.x = fmaxf(t.x, dpp_swap1(t.x));
        t.y = fmaxf(t.y, dpp_swap1(t.y));
        t.z = fmaxf(t.z, dpp_swap1(t.z));
        t.w = fmaxf(t.w, dpp_swap1(t.w));
        t += bj[j];
        m[i >> 1][j] = u32x2{pack_bf16x2(t.x, t.y), pack_bf16x2(t.z, t.w)};
      }
  }
  if (a.y == nullptr) {
#pragma unroll
    for (int i = 0; i < MT / 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) pq[i][j] = m[i][j];
  } else {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) pq[i][j] = h[i][j];
    if ((lane & 1) == 0 && px < W) {
      // both maps wanted (conv4_3, training forward): the quarter-size pooled map is stored at once
      const int Hq = H >> 1, Wq = W >> 1;
      __bf16* qimg = a.pooled + (int64_t)n * (Hq + 2) * (Wq + 2) * Cout;
#pragma unroll
      for (int i = 0; i < MT; i += 2)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (py0 + i >= H) continue;
          __bf16* dst = qimg + ((int64_t)(((py0 + i) >> 1) + 1) * (Wq + 2) + ((px >> 1) + 1)) * Cout + cbase + j * 16;
          *reinterpret_cast<u32x2*>(dst) = m[i >> 1][j];
        }
    }
  }
  }
}

// store phase, one 16-byte piece: relu on the packed pairs (rfloor = 0, or 0x80008000 = no-op without relu), then
// v_permlane16_swap pairs the 4-channel groups of lanes l and l+16
__device__ __forceinline__ u32x4 dma_store_piece(const u32x2 h0, const u32x2 h1, uint32_t rfloor) {
  u32x4 o;
  xv_pair16(u32x2{pk_max_i16(h0.x, rfloor), pk_max_i16(h0.y, rfloor)},
            u32x2{pk_max_i16(h1.x, rfloor), pk_max_i16(h1.y, rfloor)}, o);
  return o;
}

template <int WR, int WC, int MT_ = 4>
struct DmaCfg {
  static constexpr int MT = MT_;  // image rows per wave (4; 3 for the 24-row tile of configuration 22)
  static constexpr int NWAVES = WR * WC, NT = 64 * NWAVES;
  static constexpr int TH = MT * WR, TW = 16 * WC, HH = TH + 2, HW = TW + 2, NPIX = HH * HW;
  static constexpr int A_PIECES = (NPIX * 4 + 63) / 64;  // 1 KB per DMA wave-instruction
  static constexpr int A_BYTES = A_PIECES * 1024;
  static constexpr int B_PIECES = 9 * 4;  // 9 taps x (64 rows x 64 B)
  static constexpr int B_BYTES = B_PIECES * 1024;
  static constexpr int BIAS_OFF = 2 * (A_BYTES + B_BYTES);  // two 256-byte bias slots (tile parity)
  static constexpr int SK_FLAG_OFF = BIAS_OFF + 512;        // one word: "this workgroup arrived last" (stream-K tail)
  static constexpr int LDS_BYTES = SK_FLAG_OFF + 16;
  static constexpr int A_ITERS = (A_PIECES + NWAVES - 1) / NWAVES;
  static constexpr int B_ITERS = (B_PIECES + NWAVES - 1) / NWAVES;
  static_assert(LDS_BYTES <= 160 * 1024, "does not fit the LDS");
};

#ifdef XV_CLOCK_STAMP
__device__ unsigned long long xv_clk_g2[4 * XV_CLK_SLOTS];
#endif

#ifdef XV_CONV_TRACE
// debug build only (tools/conv_trace.py): per work item four cycle stamps of wave 0 of every 32nd workgroup, kept in
// spare LDS during the kernel (a global store per stamp would sit in the vmcnt queue this kernel counts on)
__device__ long long xv_trace_buf[8 * 8 * 20 * 6];
#define XV_TRACE_LDS 8192
#define XV_STAMP(k)                                                                                   \
  if ((blockIdx.x & 31) == 0 && trace_item < 20 && lane == 0)                                         \
    reinterpret_cast<long long*>(smem + C::LDS_BYTES)[(wave * 20 + trace_item) * 6 + (k)] = __builtin_readcyclecounter();
#else
#define XV_TRACE_LDS 0
#define XV_STAMP(k)
#endif

// PRIO (tuning variants, XV_DMA_PRIO): 0 = a burst raises its priority after its first MFMA and drops it at its end (default);
// 1 = static: waves 4-7 (the second-dispatched half, the arbitration loser) run at priority 1, no per-burst flips
// (MI355X_MICROARCH.md, two waves per SIMD, item 4); 2 = no priority changes at all
template <int WR, int WC, int PRIO = 0, int MT_ = 4>
__global__ __launch_bounds__(64 * WR * WC, 2) void conv_dma_kernel(ConvArgs a) {
  using C = DmaCfg<WR, WC, MT_>;
  constexpr int MT = C::MT;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WC, wc = wave % WC;
  const int l15 = lane & 15, lg = lane >> 4;
  const int H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout;
  const int Wp = W + 2;
  const int nchunks = Cin >> 5;

  const int G = gridDim.x, b = blockIdx.x;
  const int xcd = b & 7, bi = b >> 3;
  const int nb = (G - xcd + 7) >> 3;
  const int T = a.n_tiles;
  const int tq = T >> 3, trm = T & 7;
  const int t_begin = xcd * tq + (xcd < trm ? xcd : trm);
  const int t_end = t_begin + tq + (xcd < trm ? 1 : 0);
  // STREAM-K TAIL.  The nb workgroups of this XCD group walk its tiles round by round; the last round holds only
  // `rem` < nb tiles, so nb - rem CUs would idle for a whole tile (conv4_x at 16 images: 4.5 rounds, conv5_x: 1.5;
  // one image: fewer tiles than CUs on every layer from conv2 on).  With a workspace the ITEMS (tile, 32-channel
  // chunk) of that round are dealt out evenly instead: workgroup bi takes items [I bi / nb, I (bi+1) / nb) of the
  // round's I = rem * nchunks, i.e. at most two partial tiles ("units" [c0, c1) of a tile's chunks).  A partial unit
  // ends by writing its fp32 accumulators to a slab and drawing a ticket on the tile's arrival counter; the workgroup
  // that arrives LAST adds the slabs of all contributors in workgroup order (a fixed order: bitwise reproducible) and
  // runs the ordinary epilogue.  No workgroup ever waits for another one (two experts' launches share the CUs, so a
  // spin could deadlock); visibility is the agent-scope release / acquire pair of cdna_hip_programming.md, Guideline 16.
  const int Tx = t_end - t_begin;
  const int R = Tx / nb, rem = Tx - R * nb;
  const int sk_items = rem * nchunks;
  // the tail's items go to the first nbp workgroups of the group (xv_sk_parts: 0 = the exchange would cost more than
  // the idle CUs)
  const int nbp = a.sk_ws != nullptr ? xv_sk_parts(rem, nb, nchunks, C::NWAVES * MT * 4) : 0;
  const bool sk = nbp > 0;
  const int sk_a = sk && bi < nbp ? (int)((int64_t)sk_items * bi / nbp) : 0;
  const int sk_e = sk && bi < nbp ? (int)((int64_t)sk_items * (bi + 1) / nbp) : 0;
  const int nfull = sk ? R : R + (bi < rem ? 1 : 0);
  const int sk_t0 = sk_a / nchunks;  // first tail tile this workgroup touches
  const int ntail = sk_e > sk_a ? (sk_e > (sk_t0 + 1) * nchunks ? 2 : 1) : 0;
  const int nunits = nfull + ntail;
  // unit u of this workgroup: tile lid, chunks [c0, c1)
  auto unit_at = [&](int u, int& ulid, int& uc0, int& uc1) {
    if (u < nfull) {
      ulid = t_begin + bi + u * nb;
      uc0 = 0;
      uc1 = nchunks;
    } else if (u == nfull) {
      ulid = t_begin + R * nb + sk_t0;
      uc0 = sk_a - sk_t0 * nchunks;
      uc1 = sk_e - sk_t0 * nchunks < nchunks ? sk_e - sk_t0 * nchunks : nchunks;
    } else {
      ulid = t_begin + R * nb + sk_t0 + 1;
      uc0 = 0;
      uc1 = sk_e - (sk_t0 + 1) * nchunks;
    }
  };

  struct Tile {
    int n, y0, x0, co0;
  };
  auto decode = [&](int lid) {
    Tile t;
    t.co0 = (lid % a.n_ct) * 64;
    int r = lid / a.n_ct;
    t.x0 = (r % a.tiles_x) * C::TW;
    r /= a.tiles_x;
    t.y0 = (r % a.tiles_y) * C::TH;
    t.n = r / a.tiles_y;
    return t;
  };

  // per-lane source offsets (bytes, relative to the patch origin) of the patch pieces this wave moves: LDS
  // granule g = piece*64 + lane holds physical slot g&3 of pixel g>>2
  int aoff[C::A_ITERS];
#pragma unroll
  for (int it = 0; it < C::A_ITERS; ++it) {
    const int g = (wave + it * C::NWAVES) * 64 + lane;
    int p = g >> 2;
    p = p < C::NPIX ? p : C::NPIX - 1;
    const int hy = p / C::HW, hx = p - hy * C::HW;
    aoff[it] = ((hy * Wp + hx) * Cin + xv_swz32(hx, g & 3) * 8) * 2;
  }
  // LDS fragment base addresses of one item (this lane's pixel column for the three horizontal taps, its weight row):
  // functions of the lane number and the buffer parity only, but as loop invariants they (and a copy with the parity
  // added) held 8 registers the item loop does not have -- so they are re-derived for the NEXT item between taps 7 and 8,
  // when the current item's last fragment reads have been issued, under the cover of tap 8's MFMAs
  int wb, pb[3];
  auto frag_bases = [&](int bufp) {
    int ln = lane;
    asm volatile("" : "+v"(ln));  // (keeps the arithmetic from being hoisted out of the item loop)
    const int c15 = ln & 15, cg = ln >> 4;
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int hx = wc * 16 + c15 + dx;
      pb[dx] = ((wr * MT) * C::HW + hx) * 64 + (xv_swz32(hx, cg) << 4) + bufp * C::A_BYTES;
    }
    wb = 2 * C::A_BYTES + c15 * 64 + (xv_swz32(c15, cg) << 4) + bufp * C::B_BYTES;
  };
  const int64_t tap_pitch = (int64_t)nchunks * Cout * 64;  // bytes between taps of the packed image
  // channel offset of this lane's 16-byte output piece inside a 32-channel pair (see xv_pair16)
  const int csub = (lg & 1) * 16 + (lg >> 1) * 8;

  // One work item's operands = A_ITERS patch pieces + B_ITERS weight pieces per wave (1 KB each) + the tile's
  // bias.  The CU's vector-memory pipe takes ~25 cycles per piece and a wave whose DMA does not fit its queue
  // stalls in order, MFMAs included (measured: ~200 cycles per DMA when all ten were issued in one burst), so
  // the pieces are issued one patch + one weight piece per tap over the first taps of the PREVIOUS item.
  // The DMA is written in assembly: `global_load_lds` in its SGPR-base + 32-bit-VGPR-offset form keeps ONE offset
  // register per patch piece (the builtin wants a 64-bit flat pointer per lane) and keeps the instruction out of the
  // compiler's waitcnt model, which treats an LDS-DMA as a FLAT access and drains counters around it.  M0 (LDS
  // destination of the wave) is set inside the statement; nothing else in this kernel uses M0.
  auto dma16 = [&](const char* sbase, int voff, int lds_off) {
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_off), "v"(voff), "s"(sbase) : "memory");
  };
  auto dma_bases = [&](const Tile& t, int chunk, const char*& xsrc, const char*& wsrc) {
    xsrc = reinterpret_cast<const char*>(a.x) + ((((int64_t)t.n * (H + 2) + t.y0) * Wp + t.x0) * Cin + chunk * 32) * 2;
    wsrc = reinterpret_cast<const char*>(a.wpk) + (((int64_t)chunk * Cout + t.co0) << 6);
  };
  // edge tiles (the patch reaches past the padded image): patch coordinates are clamped onto the zero border, i.e.
  // the source offset is recomputed with hy <= ylim, hx <= xlim; the LDS position (and its swizzle) is unchanged
  auto dma_a = [&](const char* xsrc, int it, int buf, bool edge, int ylim, int xlim) {
    const int piece = wave + it * C::NWAVES;
    if (piece < C::A_PIECES) {
      int voff = aoff[it];
      if (edge) {
        // the clamped-offset arithmetic is re-derived from the lane number INSIDE this (rare) path: hoisted out of the
        // item loop by the compiler it held 15 registers for the whole kernel (and spilled once the loop grew)
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int g = piece * 64 + ln;
        int p = g >> 2;
        p = p < C::NPIX ? p : C::NPIX - 1;
        const int hy = p / C::HW, hx = p - hy * C::HW;
        voff = (((hy < ylim ? hy : ylim) * Wp + (hx < xlim ? hx : xlim)) * Cin + xv_swz32(hx, g & 3) * 8) * 2;
      }
      dma16(xsrc, voff, buf * C::A_BYTES + piece * 1024);
    }
  };
  auto dma_b = [&](const char* wsrc, int it, int buf) {
    const int piece = wave + it * C::NWAVES;
    if (piece < C::B_PIECES) {
      int ln = lane;
      asm volatile("" : "+v"(ln));  // (lane * 16 as a loop invariant would hold a register through the whole item loop)
      dma16(wsrc + (piece >> 2) * tap_pitch + (piece & 3) * 1024, ln * 16, 2 * C::A_BYTES + buf * C::B_BYTES + piece * 1024);
    }
  };
  // the tile's 64 bias values ride along with its first chunk (one 4-byte-per-lane DMA by the last wave)
  auto dma_bias = [&](const Tile& t, bool unit_start, int bslot) {
    if (unit_start && wave == C::NWAVES - 1)
      asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dword %1, %2" ::"s"(C::BIAS_OFF + bslot * 256), "v"(lane * 4),
                   "s"(a.bias + t.co0)
                   : "memory");
  };
  auto issue_all = [&](const Tile& t, int chunk, int buf, int bslot) {  // (first item of a workgroup: a unit start)
    const char *xsrc, *wsrc;
    dma_bases(t, chunk, xsrc, wsrc);
#pragma unroll
    for (int it = 0; it < C::A_ITERS; ++it)
      dma_a(xsrc, it, buf, t.y0 + C::TH > H || t.x0 + C::TW > W, H + 1 - t.y0, W + 1 - t.x0);
#pragma unroll
    for (int it = 0; it < C::B_ITERS; ++it) dma_b(wsrc, it, buf);
    dma_bias(t, true, bslot);
  };
  constexpr int A_TAPS = (C::A_ITERS + 1) / 2;  // taps 0..A_TAPS-1 issue the patch pieces, two each
  constexpr int LAST_DMA_TAP = A_TAPS + C::B_ITERS - 1;
  static_assert(LAST_DMA_TAP <= 8, "DMA pieces are issued inside the 9 taps");
  // full-map tile: store pieces 0 .. 2 MT - 1 leave in taps 1 .. 2 MT; those of the taps after LAST_DMA_TAP are younger
  // than every DMA
  constexpr int YTAIL = 2 * MT > LAST_DMA_TAP ? 2 * MT - LAST_DMA_TAP : 0;
  // RESIDENT WEIGHTS.  A layer with two 32-channel chunks (conv1_2, conv2_1) alternates chunk 0 / chunk 1 item by item in
  // step with the buffer parity, so weight buffer p only ever holds chunk p's weights -- of the same output-channel tile
  // too when every workgroup of the XCD keeps its tile parity (nb % n_ct == 0).  From its third item on such a workgroup
  // requests no weights at all: half the DMA pieces of these layers.  The last DMA of such an item is the bias piece in
  // tap A_TAPS (last wave, chunk 0) or the patch piece before it, so RTAIL stores are younger than every DMA.
  constexpr int RTAIL = 2 * MT > A_TAPS ? 2 * MT - A_TAPS : 0;
  static_assert(RTAIL != YTAIL && RTAIL != 2 * MT, "the counted waits must differ");
  // (a stream-K tail may start a unit at chunk 1: no fixed chunk <-> buffer parity then)
  const bool resident = nchunks == 2 && (nb % a.n_ct) == 0 && !sk && !(a.debug_same_patch & 8);  // (bit 3: XV_DMA_NO_RESIDENT, A/B timing)
  int items_done = 0;

  if (nunits == 0) return;
  int unit = 0, lid, c_first, c_end;
  unit_at(0, lid, c_first, c_end);
  if constexpr (PRIO == 1) {
    if (wave >= C::NWAVES / 2) __builtin_amdgcn_s_setprio(1);
  }
  Tile cur = decode(lid);
  int chunk = c_first, buf = 0, bslot = 0;

  f32x4 acc[MT][4];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  u32x2 pq[MT][4];
  const uint32_t rfloor = a.relu ? 0u : 0x80008000u;
  DmaPend pend{0, 0, 0, 0, false};
  const int npieces = a.y != nullptr ? 2 * MT : MT;  // 16-byte store instructions per tile and wave
  int in_flight = 0;  // stores issued after the last DMA of the previous item
  issue_all(cur, chunk, 0, 0);
  frag_bases(0);
#ifdef XV_CONV_TRACE
  int trace_item = 0;
#endif

  XV_CLK_BEGIN()
  while (true) {
    XV_STAMP(0)  // arrival at the item barrier
    // This item's operands have landed (each wave retires its own DMA; stores issued after it may stay in flight:
    // vmcnt counts in issue order), and every wave has finished reading the other buffer pair.
    if (YTAIL > 0 && in_flight == YTAIL)
      asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(YTAIL) : "memory");
    else if (RTAIL > 0 && in_flight == RTAIL)
      asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(RTAIL) : "memory");
    else if (in_flight == 2 * MT)
      asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * MT) : "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    in_flight = 0;
    XV_STAMP(1)  // barrier passed

    // Taps run dx-major: for one horizontal offset the 3 vertical taps of the wave's 4 rows touch only 6
    // patch rows, loaded once (18 + 36 fragment reads per item instead of 72).  Fragment reads run one tap
    // (weights) / one dx group (pixels) ahead in a second register set.  They are issued and waited for by
    // hand (inline asm + counted lgkmcnt): the compiler models an LDS-DMA as a FLAT access, which degrades
    // every later LDS wait of its own to lgkmcnt(0) and would drain the reads of the future with those of the
    // present.  Every wait names the fragments it releases as in/out operands, so the MFMAs that consume
    // them cannot move above it.  No other LGKM operation (s_load, ds_*) may sit inside this region -- the
    // counts below are exact: [W0 P0] W1 | W2 P1 | W3 | W4 | W5 P2 | W6 | W7 | W8 | -.
    bf16x8 wf[2][4], xf[2][6];
    static_assert(MT == 4 || MT == 3, "operand lists below");
    if constexpr (MT == 3) xf[0][5] = xf[1][5] = bf16x8{};  // named by the waits, never loaded
    constexpr int PROW = C::HW * 64;
#define XV_LDS128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
    // t = dx*3 + dy  ->  packed tap dy*3 + dx
#define XV_LDW(t, set)                                                          \
  {                                                                             \
    constexpr int tap_ = (((t) % 3) * 3 + (t) / 3) * 4096;                       \
    XV_LDS128(wf[set][0], wb, tap_);                                            \
    XV_LDS128(wf[set][1], wb, tap_ + 1024);                                     \
    XV_LDS128(wf[set][2], wb, tap_ + 2048);                                     \
    XV_LDS128(wf[set][3], wb, tap_ + 3072);                                     \
  }
#define XV_LDP(dx, set)                       \
  {                                           \
    XV_LDS128(xf[set][0], pb[dx], 0);         \
    XV_LDS128(xf[set][1], pb[dx], PROW);      \
    XV_LDS128(xf[set][2], pb[dx], 2 * PROW);  \
    XV_LDS128(xf[set][3], pb[dx], 3 * PROW);  \
    XV_LDS128(xf[set][4], pb[dx], 4 * PROW);  \
    if constexpr (MT == 4) XV_LDS128(xf[set][5], pb[dx], 5 * PROW);  \
  }
    // at most n newer reads outstanding: wf[ws] (and xf[ps]) have landed
#define XV_WAIT_W(n, ws)                                                                           \
  asm volatile("s_waitcnt lgkmcnt(%4)"                                                             \
               : "+v"(wf[ws][0]), "+v"(wf[ws][1]), "+v"(wf[ws][2]), "+v"(wf[ws][3])                 \
               : "n"(n))
#define XV_WAIT_WP(n, ws, ps)                                                                      \
  asm volatile("s_waitcnt lgkmcnt(%10)"                                                            \
               : "+v"(wf[ws][0]), "+v"(wf[ws][1]), "+v"(wf[ws][2]), "+v"(wf[ws][3]), "+v"(xf[ps][0]), \
                 "+v"(xf[ps][1]), "+v"(xf[ps][2]), "+v"(xf[ps][3]), "+v"(xf[ps][4]), "+v"(xf[ps][5])  \
               : "n"(n))
    // one 16-byte store of the previous tile per tap, taps 1..8 (piece = row p/2, channel pair p%2; pooled-only:
    // 4 pieces), each ahead of that tap's DMA: the stores younger than the last DMA are those of the taps after
    // LAST_DMA_TAP -- the count the item-end vmcnt leaves in flight
#define XV_STORE_PIECE(p)                                                                          \
  if ((p) >= 0 && (p) < 2 * MT && stores_now > (p)) {                                              \
    const u32x4 o_ = dma_store_piece(pq[(((p) >> 1) + MT) % MT][2 * ((p) & 1)],                    \
                                     pq[(((p) >> 1) + MT) % MT][2 * ((p) & 1) + 1], rfloor);       \
    if (((p) >> 1) < st_rows && st_lane) *reinterpret_cast<u32x4*>(st_ptr + ((p) & 1) * 64) = o_;  \
    if ((p) & 1) st_ptr += st_pitch;                                                               \
  }
#define XV_DMA_PIECES(t)                                                                   \
  if (has_next) {                                                                          \
    if (2 * (t) < C::A_ITERS) dma_a(nx_src, 2 * (t), buf ^ 1, nx_edge, nx_ylim, nx_xlim);         \
    if (2 * (t) + 1 < C::A_ITERS) dma_a(nx_src, 2 * (t) + 1, buf ^ 1, nx_edge, nx_ylim, nx_xlim); \
    if ((t) >= A_TAPS && (t) - A_TAPS < C::B_ITERS && !skip_b) dma_b(nw_src, (t) - A_TAPS, buf ^ 1);  \
    if ((t) == A_TAPS) dma_bias(nxt, last_chunk, bslot ^ 1);                                \
  }
#define XV_TAP(t)                                                                                  \
  {                                                                                                \
    constexpr int dx_ = (t) / 3, dy_ = (t) % 3;                                                    \
    if constexpr ((t) + 1 < 9) XV_LDW((t) + 1, ((t) + 1) & 1);                                      \
    if constexpr (dy_ == 1 && dx_ + 1 < 3) XV_LDP(dx_ + 1, (dx_ + 1) & 1);                          \
    XV_STORE_PIECE((t) - 1)                                                                        \
    XV_DMA_PIECES(t)                                                                               \
    /* reads issued after W_t: P(dx+1) of this tap (dy 1) or of the previous one (dy 2), and W_t+1 */ \
    constexpr int newer_ = ((t) + 1 < 9 ? 4 : 0) + ((dy_ != 0 && dx_ + 1 < 3) ? MT + 2 : 0);        \
    if constexpr (dy_ == 0)                                                                        \
      XV_WAIT_WP(newer_, (t) & 1, dx_ & 1);                                                        \
    else                                                                                           \
      XV_WAIT_W(newer_, (t) & 1);                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                             \
    /* Two waves share a SIMD and the older one wins every arbitration: it ran its 16-MFMA bursts back to back  \
       and let the other in only during its own gaps (measured: 144 vs ~40 MFMAs while both were active, the   \
       whole workgroup then waiting ~1.9 k cycles per item for the starved waves).  A burst therefore raises   \
       its priority after its first MFMA and drops it at its end: a wave cannot break into the other's burst, \
       and the two alternate tap by tap. */                                                                  \
    if ((t) == 0 && chunk == c_first) { /* first tap of a unit: C = 0 instead of cleared accumulators */ \
      acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[(t) & 1][0], xf[dx_ & 1][dy_], zero4, 0, 0, 0); \
      if constexpr (PRIO == 0) __builtin_amdgcn_s_setprio(2);                                      \
      __builtin_amdgcn_sched_barrier(0);                                                           \
      _Pragma("unroll") for (int i = 0; i < MT; ++i) _Pragma("unroll") for (int j = 0; j < 4; ++j) \
          if (i + j > 0)                                                                           \
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[(t) & 1][j], xf[dx_ & 1][i + dy_], zero4, 0, 0, 0); \
    } else {                                                                                       \
      acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[(t) & 1][0], xf[dx_ & 1][dy_], acc[0][0], 0, 0, 0); \
      if constexpr (PRIO == 0) __builtin_amdgcn_s_setprio(2);                                      \
      __builtin_amdgcn_sched_barrier(0);                                                           \
      _Pragma("unroll") for (int i = 0; i < MT; ++i) _Pragma("unroll") for (int j = 0; j < 4; ++j) \
          if (i + j > 0)                                                                           \
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[(t) & 1][j], xf[dx_ & 1][i + dy_], acc[i][j], 0, 0, 0); \
    }                                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                             \
    if constexpr (PRIO == 0) __builtin_amdgcn_s_setprio(1);                                        \
    __builtin_amdgcn_sched_barrier(0);                                                             \
  }
    XV_LDW(0, 0);
    XV_LDP(0, 0);
    __builtin_amdgcn_sched_barrier(0);

    const bool last_chunk = chunk + 1 == c_end;
    const bool has_next = !last_chunk || unit + 1 < nunits;
    int nlid = lid, nc_first = c_first, nc_end = c_end;
    if (last_chunk && has_next) unit_at(unit + 1, nlid, nc_first, nc_end);
    const Tile nxt = (last_chunk && has_next) ? decode(nlid) : cur;
    const int nchunk = last_chunk ? nc_first : chunk + 1;
    const char *nx_src = nullptr, *nw_src = nullptr;
    if (has_next) dma_bases(nxt, nchunk, nx_src, nw_src);
    const bool nx_edge = nxt.y0 + C::TH > H || nxt.x0 + C::TW > W;
    const int nx_ylim = H + 1 - nxt.y0, nx_xlim = W + 1 - nxt.x0;

    // the previous tile's stores go out one per tap, behind this item's DMA
    const int stores_now = pend.on ? npieces : 0;
    char* st_ptr = nullptr;  // this lane's 16-byte piece of row 0; steps one row every two pieces
    int st_pitch = 0;        // bytes
    int st_rows = 0;         // rows of the (pooled) map this wave may store: all of them except on a bottom edge tile
    bool st_lane = false;    // this lane stores: its pixel column is inside the image (and even, for the pooled map)
    bool st_edge = false;
    if (pend.on) {
      const int px = pend.x0 + wc * 16 + l15, py0 = pend.y0 + wr * MT;
      const int vrows = H - py0 < 0 ? 0 : (H - py0 < MT ? H - py0 : MT);
      st_edge = pend.y0 + C::TH > H || pend.x0 + C::TW > W;
      if (a.y != nullptr) {
        st_pitch = Wp * Cout * 2;
        st_rows = vrows;
        st_lane = px < W;
        st_ptr = reinterpret_cast<char*>(a.y + (int64_t)pend.n * (H + 2) * Wp * Cout +
                                         ((int64_t)(py0 + 1) * Wp + (px + 1)) * Cout + pend.co0 + csub);
      } else {
        const int Hq = H >> 1, Wq = W >> 1;
        st_pitch = (Wq + 2) * Cout * 2;
        st_rows = (vrows + 1) >> 1;
        st_lane = px < W && (lane & 1) == 0;
        st_ptr = reinterpret_cast<char*>(a.pooled + (int64_t)pend.n * (Hq + 2) * (Wq + 2) * Cout +
                                         ((int64_t)((py0 >> 1) + 1) * (Wq + 2) + ((px >> 1) + 1)) * Cout + pend.co0 + csub);
      }
      pend.on = false;
    }
    // stores issued after the last DMA piece of this item (pieces LAST_DMA_TAP .. npieces-1 go out in later taps)
    // (an edge tile may skip store instructions: no counted wait then)
    const bool skip_b = resident && items_done >= 1;  // the NEXT item is this workgroup's third or later
    in_flight = (stores_now == 2 * MT && !st_edge) ? (skip_b ? RTAIL : YTAIL) : 0;
    __builtin_amdgcn_sched_barrier(0);
    XV_STAMP(2)  // first fragments requested, DMA issued

    XV_TAP(0) XV_TAP(1) XV_TAP(2) XV_TAP(3) XV_TAP(4) XV_TAP(5) XV_TAP(6) XV_TAP(7)
    frag_bases(buf ^ 1);  // (tap 8 computes from registers: the bases of the current buffers are dead)
    __builtin_amdgcn_sched_barrier(0);
    XV_TAP(8)
    XV_STAMP(3)  // all MFMAs of the item issued

    if (last_chunk) {
      bool finish = true;  // this workgroup holds the tile's complete sums
      if (c_first > 0 || c_end < nchunks) {
        // ---- a partial unit of the stream-K tail ----
        // Hand-off form (cdna_hip_programming.md, Guideline 16, "every load sc1"; MI355X_MICROARCH.md, Valid forms, first
        // row): slabs are stored WRITE-THROUGH (sc1) and read by sc1 loads, every storing wave drains its stores before
        // the workgroup's barrier, ONE lane then adds to the tile's arrival counter (agent-scope atomic) and the workgroup
        // whose add came last -- told by the value the add returned -- reads the slabs behind a barrier that lane joins.
        // No release / acquire fence: a release would write back the whole L2 of the XCD (megabytes of freshly stored
        // output) in every partial workgroup -- measured 30-40 us per launch.
        const int r = lid - (t_begin + R * nb);  // tail tile of this XCD group
        const int slot = unit > nfull ? 1 : 0;   // the workgroup's first / second tail unit
        // contributors of tile r: the workgroups whose (non-empty: fewer items than workgroups leaves some without any)
        // item range meets [r nchunks, (r+1) nchunks)
        auto contributes = [&](int w, int& ws) {
          ws = (int)((int64_t)sk_items * w / nbp);
          const int we = (int)((int64_t)sk_items * (w + 1) / nbp);
          return we > ws && we > r * nchunks && ws < (r + 1) * nchunks;
        };
        int ncontrib = 0;
        for (int w = 0, ws; w < nbp; ++w) ncontrib += contributes(w, ws) ? 1 : 0;
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(a.sk_ws, 0, XV_SK_HDR + 2 * (int)gridDim.x * XV_SK_SLAB, 0x00020000);
        const int lane_off = wave * (MT * 4 * 1024) + lane * 16;  // this lane's first 16 bytes inside a slab
        unsigned* cnt = reinterpret_cast<unsigned*>(a.sk_ws) + xcd * 64 + r;
        volatile int* flag = reinterpret_cast<volatile int*>(smem + C::SK_FLAG_OFF);
        // (1) a tile split in TWO (the tail of an even split at 16 images): the other half already here?  Then this
        // workgroup is the last arriver for certain (nobody is left to add) and publishes nothing -- a + b == b + a bit
        // for bit, so its own sums can stay in registers whichever half arrives last
        finish = false;
        if (ncontrib == 2) {
          if (tid == 0) *flag = __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 1u;
          asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
          finish = __builtin_amdgcn_readfirstlane(*flag) != 0;  // (wave-uniform by construction: keep the branch scalar)
        }
        if (!finish) {
          // (2) publish the partial sums write-through, drain, barrier, ONE lane draws the ticket
          const int mine = XV_SK_HDR + (b * 2 + slot) * XV_SK_SLAB + lane_off;
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][j]), rsrc, mine + (i * 4 + j) * 1024, 0, 16);
          asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");  // (the flag word has been read by all)
          if (tid == 0)
            *flag = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 == (unsigned)ncontrib;
          asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
          finish = __builtin_amdgcn_readfirstlane(*flag) != 0;
        }
        if (finish) {
          // last arriver: the tile = the contributors' partial sums added in workgroup order (two halves: its own from
          // registers, see above; more: all of them, its own included, from the slabs -- one fixed expression whoever
          // arrives last)
          if (tid == 0) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // zero for the next launch
          bool first_slab = ncontrib != 2;
          for (int w = 0; w < nbp; ++w) {
            int ws;
            if (!contributes(w, ws) || (ncontrib == 2 && w == bi)) continue;
            const int theirs = XV_SK_HDR + ((w * 8 + xcd) * 2 + (ws < r * nchunks ? 1 : 0)) * XV_SK_SLAB + lane_off;
            u32x4 v[MT][4];  // the whole slab share of this lane in flight before the first add (one memory round trip)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
              for (int j = 0; j < 4; ++j) v[i][j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, theirs + (i * 4 + j) * 1024, 0, 16);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
              for (int j = 0; j < 4; ++j)
                acc[i][j] = first_slab ? __builtin_bit_cast(f32x4, v[i][j]) : acc[i][j] + __builtin_bit_cast(f32x4, v[i][j]);
            first_slab = false;
          }
        }
        in_flight = 0;
      }
      if (finish) {
        dma_epilogue_pack<MT>(a, acc, pq, reinterpret_cast<const float*>(smem + C::BIAS_OFF + bslot * 256), cur.n,
                              cur.y0 + wr * MT, cur.x0 + wc * 16 + l15, cur.co0 + lg * 4, lane);
        pend = DmaPend{cur.n, cur.y0, cur.x0, cur.co0, true};
        // both maps wanted: the pack phase stored the pooled map (MT/2 x 4 instructions), the youngest operations now
        static_assert(YTAIL != 2 * MT, "the two counted waits must differ");
        in_flight = (a.y != nullptr && a.pooled != nullptr && !(cur.y0 + C::TH > H || cur.x0 + C::TW > W)) ? 2 * MT : 0;
      }
      XV_STAMP(4)  // tile converted
      bslot ^= 1;
    }
#ifdef XV_CONV_TRACE
    ++trace_item;
#endif
    if (!has_next) {
      if (!pend.on) break;  // (the last unit was a partial one that another workgroup completes)
      // last tile of this workgroup: store it now
      const int px = pend.x0 + wc * 16 + l15, py0 = pend.y0 + wr * MT;
      const int vrows = H - py0 < 0 ? 0 : (H - py0 < MT ? H - py0 : MT);
      if (a.y != nullptr) {
        __bf16* base = a.y + (int64_t)pend.n * (H + 2) * Wp * Cout + ((int64_t)(py0 + 1) * Wp + (px + 1)) * Cout + pend.co0 + csub;
#pragma unroll
        for (int t = 0; t < 2 * MT; ++t) {
          const u32x4 o = dma_store_piece(pq[t >> 1][2 * (t & 1)], pq[t >> 1][2 * (t & 1) + 1], rfloor);
          if ((t >> 1) < vrows && px < W) *reinterpret_cast<u32x4*>(base + (int64_t)(t >> 1) * Wp * Cout + (t & 1) * 32) = o;
        }
      } else {
        const int Hq = H >> 1, Wq = W >> 1;
        __bf16* base = a.pooled + (int64_t)pend.n * (Hq + 2) * (Wq + 2) * Cout +
                       ((int64_t)((py0 >> 1) + 1) * (Wq + 2) + ((px >> 1) + 1)) * Cout + pend.co0 + csub;
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          const u32x4 o = dma_store_piece(pq[t >> 1][2 * (t & 1)], pq[t >> 1][2 * (t & 1) + 1], rfloor);
          if ((t >> 1) < ((vrows + 1) >> 1) && px < W && (lane & 1) == 0)
            *reinterpret_cast<u32x4*>(base + (int64_t)(t >> 1) * (Wq + 2) * Cout + (t & 1) * 32) = o;
        }
      }
      break;
    }
    if (last_chunk) {
      ++unit;
      c_first = nc_first;
      c_end = nc_end;
    }
    lid = nlid;
    cur = nxt;
    chunk = nchunk;
    buf ^= 1;
    ++items_done;
  }
  XV_CLK_END(xv_clk_g2)
#ifdef XV_CONV_TRACE
  __syncthreads();
  if (wave == 0 && (blockIdx.x & 31) == 0)
    for (int i = lane; i < 8 * 20 * 6; i += 64)
      xv_trace_buf[(blockIdx.x >> 5) * 960 + i] = reinterpret_cast<long long*>(smem + C::LDS_BYTES)[i];
#endif
#undef XV_TAP
#undef XV_STORE_PIECE
#undef XV_DMA_PIECES
#undef XV_WAIT_WP
#undef XV_WAIT_W
#undef XV_LDP
#undef XV_LDW
#undef XV_LDS128
}

template <int WR, int WC, int PRIO = 0, int MT = 4>
int launch_conv_dma(const ConvArgs& a0, hipStream_t stream) {
  using C = DmaCfg<WR, WC, MT>;
  if (MT % 2 != 0 && a0.pooled != nullptr) return XV_ESHAPE;  // the fused pool pairs rows inside a wave
  ConvArgs a = a0;
  static const bool no_resident = getenv("XV_DMA_NO_RESIDENT") != nullptr;
  if (no_resident) a.debug_same_patch |= 8;
  // second half of the packed buffer: the 32-channel-chunk image
  a.wpk = a0.wpk + (int64_t)9 * a.Cin * a.Cout;
  a.tiles_x = (a.W + C::TW - 1) / C::TW;
  a.tiles_y = (a.H + C::TH - 1) / C::TH;
  a.n_ct = a.Cout / 64;
  static bool attr_set[XV_MAX_DEVICES] = {false};
  {
    const hipError_t e = xv_allow_dynamic_lds(reinterpret_cast<const void*>(&conv_dma_kernel<WR, WC, PRIO, MT>),
                                              C::LDS_BYTES + XV_TRACE_LDS, attr_set);
    if (e != hipSuccess) return (int)e;
  }
  const int64_t ntiles = (int64_t)a.tiles_x * a.tiles_y * a.N * a.n_ct;
  if (ntiles <= 0 || ntiles > 0x7fffffff) return XV_ESHAPE;
  if ((int64_t)(C::HH * (a.W + 2) + C::HW) * a.Cin * 2 > 0x7fffffff) return XV_ESHAPE;  // 32-bit patch offsets
  a.n_tiles = (int)ntiles;
  const int64_t slots = a.num_cus;
  static const bool no_sk = getenv("XV_DMA_NO_STREAMK") != nullptr;  // (A/B timing)
  if (no_sk || a.Cin < 64 || (slots & 7) != 0 || slots > XV_SK_MAX_CUS) a.sk_ws = nullptr;
  // with a stream-K workspace every CU gets a workgroup: the kernel deals the items of an incomplete round out evenly
  const int64_t nblk = a.sk_ws != nullptr ? slots : (ntiles < slots ? ntiles : slots);
  static const bool grid_only = getenv("XV_SK_GRID_ONLY") != nullptr;  // (experiment: the full grid without the tail split)
  if (grid_only) a.sk_ws = nullptr;
  hipLaunchKernelGGL((conv_dma_kernel<WR, WC, PRIO, MT>), dim3((unsigned)nblk), dim3(C::NT), C::LDS_BYTES + XV_TRACE_LDS, stream, a);
  return xv_launch_status();
}

// ---- weight packing: fp32 HWIO -> bf16 [tap][cin/64][cout][64] (16-byte slots swizzled by xv_swz), followed for
// 3x3 filters by the generation-2 image [tap][cin/32][cout][32] (xv_swz32).  dgrad != 0 packs the weights of the
// data-gradient convolution instead: input/output channels swapped and the taps point-reflected.
__device__ __forceinline__ void pack_weights_image(const float* __restrict__ w, __bf16* __restrict__ out, int taps,
                                                   int cin, int cout, int dgrad, int64_t idx, int64_t total) {
  // logical operand of the convolution that will consume the image: Wl[tap][ci][co]; for the data gradient
  // Wl[tap][ci = dgrad input = cout of w][co = cin of w] = w[taps-1-tap][co][ci]
  const int rc = dgrad ? cout : cin;  // reduction channels
  const int oc = dgrad ? cin : cout;  // output channels
  auto wl = [&](int tap, int ci, int co) -> float {
    return dgrad ? w[((int64_t)(taps - 1 - tap) * cin + co) * cout + ci] : w[((int64_t)tap * cin + ci) * cout + co];
  };
  const int nch64 = rc >> 6, nch32 = rc >> 5;
  {  // image 1, destination-linear [tap][chunk64][co][phys_slot 0..7][e]
    const int e = (int)(idx & 7);
    const int ps = (int)((idx >> 3) & 7);
    int64_t rest = idx >> 6;
    const int co = (int)(rest % oc);
    rest /= oc;
    const int chunk = (int)(rest % nch64);
    const int tap = (int)(rest / nch64);
    const int sl = xv_swz(co, ps);  // involution: logical slot stored at this physical slot
    out[idx] = (__bf16)wl(tap, chunk * 64 + sl * 8 + e, co);
  }
  if (taps == 9) {  // image 2 (generation-2 kernel), [tap][chunk32][co][phys_slot 0..3][e]
    const int e = (int)(idx & 7);
    const int ps = (int)((idx >> 3) & 3);
    int64_t rest = idx >> 5;
    const int co = (int)(rest % oc);
    rest /= oc;
    const int chunk = (int)(rest % nch32);
    const int tap = (int)(rest / nch32);
    const int sl = xv_swz32(co, ps);
    out[total + idx] = (__bf16)wl(tap, chunk * 32 + sl * 8 + e, co);
    // image 3 (generations 4 / 5 on 16x16x32 MFMA blocks, configurations 26-28, and the fused first pair): rows permuted inside every 64-row block (row
    // 16 j + 4 g + q = channel 16 g + 4 j + q: an involution), slots swizzled by (rho >> 1) & 2
    const int m6 = co & 63;
    const int ch16 = (co & ~63) + 16 * ((m6 >> 2) & 3) + 4 * (m6 >> 4) + (m6 & 3);
    out[2 * total + idx] = (__bf16)wl(tap, chunk * 32 + (ps ^ ((co >> 1) & 2)) * 8 + e, ch16);
  }
}

// out: the image selected by `dgrad`; out_dgrad (may be null): additionally the data-gradient image, so that a
// training step re-packs a layer's forward and backward weights in one launch
__global__ void pack_weights_kernel(const float* __restrict__ w, __bf16* __restrict__ out, int taps, int cin,
                                    int cout, int dgrad, __bf16* __restrict__ out_dgrad) {
  const int64_t total = (int64_t)taps * cin * cout;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    pack_weights_image(w, out, taps, cin, cout, dgrad, idx, total);
    if (out_dgrad != nullptr) pack_weights_image(w, out_dgrad, taps, cin, cout, 1, idx, total);
  }
}

// Every layer of a model in ONE launch (the optimizer step re-packs all kernels: 18 launches of ~12 us each were 2 % of a
// training step): blockIdx.y = table entry, blockIdx.x strides over that kernel's 8-channel groups.  One thread = the 8
// reduction channels of one logical 16-byte slot for one (tap, output channel): 8 floats in (lanes = consecutive output
// channels: coalesced for the forward image, 32 contiguous bytes per lane for the data-gradient image), the same 16 bytes
// out to both images -- the element-per-thread form (pack_weights_image, kept for the single-layer entry points: same
// bits) gathered 2 bytes at a time and took 210 us per training step for 14.7 M weights.
__device__ __forceinline__ void pack_weights_group(const float* __restrict__ w, __bf16* __restrict__ out, int taps, int cin,
                                                   int cout, int dgrad, int64_t g, int64_t total) {
  const int rc = dgrad ? cout : cin;  // reduction channels
  const int oc = dgrad ? cin : cout;  // output channels
  // g -> (tap, 64-channel chunk, output channel, slot of the chunk), the slot fastest: the eight lanes of one output channel
  // fill one 128-byte row of image 1 (two 64-byte rows of images 2 / 3), eight consecutive channels per wave a contiguous
  // 1 KB -- with the output channel fastest (round 3) every lane wrote its 16 bytes into a row of its own and the re-pack
  // took 167 us of a training step for 235 MB of traffic.  The data-gradient image reads 8 x 32 contiguous bytes the same way.
  const int nc64 = rc >> 6;          // rc is a multiple of 64: every packed layout is made of 64-channel rows
  const int64_t r = g >> 3;
  const int co = (int)(r % oc);
  const int64_t rest = r / oc;
  const int G = (int)(rest % nc64) * 8 + (int)(g & 7);
  const int tap = (int)(rest / nc64);
  float v[8];
  if (dgrad) {
    const float* src = w + ((int64_t)(taps - 1 - tap) * cin + co) * cout + G * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = src[e];
  } else {
    const float* src = w + ((int64_t)tap * cin + G * 8) * cout + co;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = src[(int64_t)e * cout];
  }
  const u32x4 o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
  {  // image 1: [tap][chunk64][co][phys_slot 0..7][e]
    const int64_t at = ((((int64_t)tap * (rc >> 6) + (G >> 3)) * oc + co) << 6) + xv_swz(co, G & 7) * 8;
    *reinterpret_cast<u32x4*>(out + at) = o;
  }
  if (taps == 9) {  // image 2 (generation-2 kernel): [tap][chunk32][co][phys_slot 0..3][e]
    const int64_t at = ((((int64_t)tap * (rc >> 5) + (G >> 2)) * oc + co) << 5) + xv_swz32(co, G & 3) * 8;
    *reinterpret_cast<u32x4*>(out + total + at) = o;
    // image 3 (generations 4 / 5, 16x16x32 blocks): row 16 j + 4 g + q of the 64-row block for channel 16 g + 4 j + q
    const int c6 = co & 63;
    const int rho16 = (co & ~63) + 16 * ((c6 >> 2) & 3) + 4 * (c6 >> 4) + (c6 & 3);
    const int64_t at4 = ((((int64_t)tap * (rc >> 5) + (G >> 2)) * oc + rho16) << 5) + (((G & 3) ^ ((rho16 >> 1) & 2)) << 3);
    *reinterpret_cast<u32x4*>(out + 2 * total + at4) = o;
  }
}

__global__ void pack_weights_multi_kernel(const xv_pack_desc* __restrict__ table) {
  const xv_pack_desc d = table[blockIdx.y];
  const int taps = d.k * d.k;
  const int64_t total = (int64_t)taps * d.cin * d.cout;
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < (total >> 3); g += (int64_t)gridDim.x * blockDim.x) {
    pack_weights_group(d.w_hwio, reinterpret_cast<__bf16*>(d.packed), taps, d.cin, d.cout, 0, g, total);
    if (d.packed_dgrad != nullptr)
      pack_weights_group(d.w_hwio, reinterpret_cast<__bf16*>(d.packed_dgrad), taps, d.cin, d.cout, 1, g, total);
  }
}

__global__ void pack_weights_f8_header_kernel(char* __restrict__ out, int scale_exp) {
  reinterpret_cast<int*>(out)[threadIdx.x] = threadIdx.x == 0 ? scale_exp : 0;
}

// fp8 image: [256-byte header][tap][cin/128][cout][128 B, 16-byte slots swizzled by xv_swz]; one thread = 4 bytes
__global__ void pack_weights_f8_kernel(const float* __restrict__ w, char* __restrict__ out, int taps, int cin, int cout,
                                       int scale_exp, float mul) {
  const int64_t total4 = (int64_t)taps * cin * cout / 4;
  const int nch = cin >> 7;
  if (blockIdx.x == 0 && threadIdx.x < 64) reinterpret_cast<int*>(out)[threadIdx.x] = threadIdx.x == 0 ? scale_exp : 0;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total4; q += (int64_t)gridDim.x * blockDim.x) {
    const int64_t idx = q * 4;  // destination byte
    const int e = (int)(idx & 15);
    const int ps = (int)((idx >> 4) & 7);
    int64_t rest = idx >> 7;
    const int co = (int)(rest % cout);
    rest /= cout;
    const int chunk = (int)(rest % nch);
    const int tap = (int)(rest / nch);
    const int ci = chunk * 128 + xv_swz(co, ps) * 16 + e;
    f32x4 v;
    v.x = w[((int64_t)tap * cin + ci) * cout + co];
    v.y = w[((int64_t)tap * cin + ci + 1) * cout + co];
    v.z = w[((int64_t)tap * cin + ci + 2) * cout + co];
    v.w = w[((int64_t)tap * cin + ci + 3) * cout + co];
    *reinterpret_cast<uint32_t*>(out + 256 + idx) = pack_fp8x4(v, mul);
  }
}

// ---- tile configurations ---------------------------------------------------------------------
// id: pixel patch x output channels, waves, LDS, workgroups per CU (register bound)
//   0: 16x16 x 128, 4 waves, 74 KB, 2/CU          1:  8x16 x 128, 4 waves, 55 KB, 2/CU
//   2:  8x32 x 128, 4 waves, 75 KB, 2/CU          3: 16x32 x 128, 8 waves, 110 KB, 1/CU
//   4: 16x16 x  64, 4 waves, 57 KB, 2/CU          5: 16x32 x  64, 4 waves, 94 KB, 1/CU
//   6:  8x32 x  64, 4 waves, 59 KB, 2/CU          7:  8x16 x 256, 4 waves, 87 KB, 1/CU
//   8: as 0 but 1/CU with the patch prefetch      9: as 2 but 1/CU with the patch prefetch
//  10: as 4 with two taps per barrier (74 KB)    11: as 6 with two taps per barrier (76 KB)
//  12: 16x32 x 64, 8 waves, 3 taps per barrier, 126 KB, 1/CU
//  13: 16x32 x 64, 8 waves, 5 taps per barrier, 158 KB, 1/CU
//  14 / 15 / 16: as 10 / 11 / 13 with the weight tiles staged by LDS-DMA
//  17: generation 2 (conv_dma_kernel): 16x32 x 64, 8 waves, 32-channel chunks, all operands by LDS-DMA, 150 KB, 1/CU;
//      3x3 only
// Tried and dropped (slower, tools/conv_tune.py): 8-wave 128-channel tiles with 2-3 taps per barrier,
// a single-weight-buffer variant at three workgroups per CU, two-wave workgroups at four per CU,
// weight fragments streamed L1 -> VGPR without LDS, s_setprio around the MFMA clusters.
//  18: generation 3 for 1x1 convs (conv1x1_gemm.hip): flat GEMM over the padded rows, 128 px x 128 channels, 4 waves,
//      both operands by LDS-DMA, 64 KB, 2/CU
//  19 / 20: 16x16 / 8x32 x 128 output channels, 8 waves (4 pixel waves x 2 channel halves), two taps per barrier, weight
//      stages by LDS-DMA, 105 KB, 1/CU: the staged patch is shared by both channel halves (half the patch traffic per
//      FLOP of configurations 14 / 15) -- built for the fp8 kernel, which is bound by what feeds it
//  21: generation 2b (conv_dma2_kernel): as 17 with the item barrier and the next item's first fragment reads inside tap 8
//  22: generation 2 on a 24x16 tile (8 waves x 3 rows x 16 columns; 132 KB): the 24x48 conv5 maps of a 768x384 input, which
//      the 16x32 tile covers at 56 %; no fused pool (a wave holds an odd number of rows)
//  23: the narrow flat GEMM for 1x1 convs onto 64 channels (conv1x1_gemm.hip): 64 padded rows x 64 channels, 4 waves,
//      32 KB, several workgroups per CU: the FCN's two score convs
//  24: generation 4 (conv_f8_dma.hip): e4m3 in and out, 16x32 x 64, 8 waves, 64-channel chunks on
//      v_mfma_scale_f32_32x32x64_f8f6f4, all operands by LDS-DMA, 151 KB, 1/CU; maps that tile exactly
//  25: (retired in round 4: generation 4 on v_mfma_f32_32x32x16_bf16; held 0.2 GHz less clock than 26 on real data)
//  26: generation 4 on bf16 operands, v_mfma_f32_16x16x32_bf16, 32-channel chunks: bias + relu (+ pool), data-gradient epilogue
//  27 / 28: generation 5 (conv_col_dma.hip): the loop of 26 on a column of 8 waves x 3 / 4 rows x 16 columns -- 24x16 and
//      32x16 tiles (the 24x48 conv5 maps tile exactly in 24x16); exact tilings only, fused pool on 28 only
constexpr int XV_NUM_CONV_CFG = 29;
struct Geo {
  int th, tw, bn, per_cu;
};
const Geo kGeo[XV_NUM_CONV_CFG] = {{16, 16, 128, 2}, {8, 16, 128, 2}, {8, 32, 128, 2}, {16, 32, 128, 1},
                                   {16, 16, 64, 2},  {16, 32, 64, 1}, {8, 32, 64, 2},  {8, 16, 256, 1},
                                   {16, 16, 128, 1}, {8, 32, 128, 1}, {16, 16, 64, 2},  {8, 32, 64, 2},
                                   {16, 32, 64, 1},  {16, 32, 64, 1},  {16, 16, 64, 2},  {8, 32, 64, 2},
                                   {16, 32, 64, 1},  {16, 32, 64, 1},  {1, 128, 128, 2},
                                   {16, 16, 128, 1}, {8, 32, 128, 1}, {16, 32, 64, 1}, {24, 16, 64, 1}, {1, 64, 64, 4},
                                   {16, 32, 64, 1},  {16, 32, 64, 1},  {16, 32, 64, 1},  {24, 16, 64, 1}, {32, 16, 64, 1}};

template <int KS>
int launch_cfg(int cfg, const ConvArgs& a, hipStream_t s) {
  if (cfg < 0 || cfg >= XV_NUM_CONV_CFG) return XV_EINVAL;
  if (a.Cout % kGeo[cfg].bn) return XV_ESHAPE;
  if (cfg == 24) {
    if (KS != 3 || !a.in_f8 || !a.out_f8 || a.mask != nullptr || a.addend != nullptr) return XV_ESHAPE;
    return xv_launch_conv3x3_f8_dma(a.x, a.wpk, a.bias, a.y, a.pooled, a.N, a.H, a.W, a.Cin, a.Cout, a.relu, 1, 1, a.scale_x,
                                    a.out_mul, a.num_cus, s);
  }
  if (cfg == 25) return XV_ESHAPE;  // (retired: generation 4 on 32x32x16 bf16 blocks, superseded by 26; its weight image is gone)
  // Retired in round 5 (never chosen by pick_cfg, the stream-K tail, a partial-tile fallback or the fp8 path): the
  // first-generation tile variants 0, 2-13, 19, 20 and generation 2b (21).  Their indices stay reserved.
  if (cfg == 0 || (cfg >= 2 && cfg <= 13) || cfg == 19 || cfg == 20 || cfg == 21) return XV_ESHAPE;
  if (cfg == 26) {
    if (KS != 3 || a.in_f8) return XV_ESHAPE;
    return xv_launch_conv3x3_f8_dma(a.x, a.wpk, a.bias, a.y, a.pooled, a.N, a.H, a.W, a.Cin, a.Cout, a.relu, 0, a.out_f8, 0,
                                    a.out_mul, a.num_cus, s, nullptr, 1, a.mask, a.addend);
  }
  if (cfg == 27 || cfg == 28) {
    if (KS != 3 || a.in_f8 || a.out_f8) return XV_ESHAPE;
    return xv_launch_conv3x3_col(a.x, a.wpk, a.bias, a.y, a.pooled, a.N, a.H, a.W, a.Cin, a.Cout, a.relu, cfg == 27 ? 3 : 4,
                                 a.num_cus, s, a.mask, a.addend, a.split_ws, a.split_bytes);
  }
  if (a.in_f8) {
    // the fp8 kernel is built for the tile shapes with LDS-DMA weight stages (the ones the bf16 chooser falls back to)
    switch (cfg) {
      case 14: return launch_conv<4, 4, 1, 1, KS, 2, 2, 1, 1>(a, s);
      case 15: return launch_conv<4, 2, 2, 1, KS, 2, 2, 1, 1>(a, s);
      case 16: return launch_conv<4, 4, 2, 1, KS, 2, 5, 1, 1>(a, s);
      default: return XV_ESHAPE;
    }
  }
  if (a.out_f8 && (cfg == 17 || cfg == 18 || cfg == 21 || cfg == 22 || cfg == 23)) return XV_ESHAPE;  // fp8 outputs come from the shared first-generation epilogue
  switch (cfg) {
    case 1: return launch_conv<4, 2, 1, 2, KS, 2>(a, s);
    case 14: return launch_conv<4, 4, 1, 1, KS, 2, 2, 1>(a, s);
    case 15: return launch_conv<4, 2, 2, 1, KS, 2, 2, 1>(a, s);
    case 16: return launch_conv<4, 4, 2, 1, KS, 2, 5, 1>(a, s);
    case 22:
      if constexpr (KS == 3) return launch_conv_dma<8, 1, 0, 3>(a, s);
      return XV_ESHAPE;
    case 23:
      if constexpr (KS == 1) {
        if (a.pooled != nullptr || a.y == nullptr) return XV_ESHAPE;
        return xv_launch_conv1x1_n64(a.x, a.wpk, a.bias, a.y, a.mask, a.addend, a.N, a.H, a.W, a.Cin, a.Cout, a.relu, s);
      }
      return XV_ESHAPE;
    case 17:
      if constexpr (KS == 3) {
        return launch_conv_dma<4, 2>(a, s);
      }
      return XV_ESHAPE;
    default:
      if constexpr (KS == 1) {
        if (a.pooled != nullptr || a.y == nullptr) return XV_ESHAPE;
        return xv_launch_conv1x1_gemm(a.x, a.wpk, a.bias, a.y, a.mask, a.addend, a.N, a.H, a.W, a.Cin, a.Cout, a.relu, s);
      }
      return XV_ESHAPE;
  }
}

// Default choice, from tools/conv_tune.py on MI355X (profiles/r1_conv_tune_b8.txt): 64 output
// channels x 4 rows per wave, two taps per barrier, weight tiles by LDS-DMA.  Large images (conv1_2 / conv2_x at 8+ images)
// prefer the 8-wave 16x32 patch with five taps per barrier (lower halo + barrier overhead); otherwise
// two 4-wave workgroups per CU, patch shape by least waste on partial tiles.
int pick_cfg(const ConvArgs& a, int k) {
  auto covered = [&](int c) {
    const Geo& g = kGeo[c];
    return (double)((a.H + g.th - 1) / g.th * g.th) * ((a.W + g.tw - 1) / g.tw * g.tw);
  };
  const double g1 = covered(14) < covered(15) ? covered(14) : covered(15);
  if (a.in_f8 || a.out_f8) {
    // generation 4 wherever the map tiles exactly in 16x32 (XV_F8_NO_GEN4=1: first generation everywhere, A/B timing)
    static const bool no_gen4 = getenv("XV_F8_NO_GEN4") != nullptr;
    // (partial tiles: only where the 16x32 tile covers the map well enough -- or where nothing else exists: 64-channel
    // e4m3 chunks)
    if (k == 3 && a.in_f8 && a.out_f8 && !no_gen4 && xv_conv3x3_f8_dma_ok(a.H, a.W, a.Cin, a.Cout) &&
        (xv_conv3x3_dma4_exact(a.H, a.W) || (a.Cin & 127) || covered(16) <= 1.3 * g1))
      return 24;
    // the bf16 conv that writes the first e4m3 map: generation 2 has no e4m3 epilogue, generation 4 does
    if (k == 3 && !a.in_f8 && a.out_f8 && !no_gen4 && xv_conv3x3_dma4_bf16_ok(a.H, a.W, a.Cin, a.Cout)) return 26;
    if (a.in_f8 && (a.Cin & 127)) return -1;  // 64-channel e4m3 chunks: generation 4 only
    // 16x32 patch, 8 waves, five taps per barrier where it tiles the map (large maps); else two 4-wave workgroups
    if (k == 3 && covered(16) <= g1 && (int64_t)a.N * a.H * a.W >= XV_F8_BIG_MAP) return 16;
    return covered(15) < covered(14) ? 15 : 14;
  }
  // Generation 2 unless its partial tiles waste more than its per-pixel advantage over the 16x16 / 8x32 tiles of
  // generation 1 (~1.25x on 16x32 tiles, ~1.15x on the 24x16 tile of configuration 22; conv_tune.py at 16 images: conv5_1
  // 1 064 against 910 TFLOP/s).  Between its two tiles the ROUNDS decide (one workgroup per CU, so a launch takes
  // ceil(items / CUs) item times): the 24x48 conv5 maps tile exactly in 24x16 (2 rounds of 384-pixel items against 2 of
  // 512), the 48x96 conv4 maps at 16 images make 6 even rounds instead of 4.5, and at one or two images -- fewer items
  // than CUs -- the smaller item simply ends sooner.  No fused pool on the 24x16 tile.
  if (k == 3) {
    auto round_cost = [&](int c, double speed) {
      const Geo& g = kGeo[c];
      const int64_t items = (int64_t)a.N * ((a.H + g.th - 1) / g.th) * ((a.W + g.tw - 1) / g.tw) * (a.Cout / 64);
      double rounds = (double)((items + a.num_cus - 1) / a.num_cus);
      // From two rounds of workgroups on, the WORK decides, not the round count: the experts of a fusion model run on two
      // streams and the other expert's workgroups fill an incomplete last round (conv4_x at 16 images: 4.5 rounds of
      // 16x32 tiles against 6 even rounds of the slower 24x16 tile -- the same time one expert alone, 1.5-3.6 % of the
      // whole two-stream step in favour of the 16x32 tile; XV_CFG_ROUNDS=1 restores the round count for A/B timing).
      // Below two rounds a launch is latency: the rounds decide.
      static const bool by_rounds = getenv("XV_CFG_ROUNDS") != nullptr;
      if (!by_rounds && items >= 2 * (int64_t)a.num_cus) rounds = (double)items / a.num_cus;
      const int nchunks = a.Cin / 32;
      if (a.sk_ws != nullptr && items % a.num_cus && (a.num_cus & 7) == 0) {
        // stream-K tail (conv_dma_kernel): the last round's items dealt out over nbp workgroups per XCD group
        const int nb = a.num_cus / 8, rem = (int)((items % a.num_cus + 7) / 8);
        const int nbp = xv_sk_parts(rem < nb ? rem : nb - 1, nb, nchunks, g.th == 24 ? 96 : 128);
        if (nbp > 0)
          rounds = (double)(items / a.num_cus) + (double)((rem * nchunks + nbp - 1) / nbp) / nchunks +
                   (nbp * 128 / 125 + 8) / (2.9 * nchunks);
      }
      return rounds * g.th * g.tw / speed;
    };
    // generation 4 where the map tiles exactly in 16x32: the leaner item loop on v_mfma_f32_16x16x32_bf16 (configuration 26).
    // Round 4 measured the clock inside the item loops (profiles/r4_conv_inkernel_clock.json): on random operands the
    // 32x32x16 form (25) holds 1.69-1.76 GHz, this one 1.89-1.98 GHz at 3 % more cycles, generation 2 (17: 16x16x32 too, a
    // 15 % longer loop) 1.99-2.12 GHz; on zeros all three run at 2.39 GHz.  26 is ahead of both on every layer shape
    // (profiles/r4_conv_mfma_shape_ab.txt: +7-10 % on conv3_x / conv4_x, +2-8 % on the one- and two-chunk layers).
    // XV_BF16_GEN4=0: never; =2: only from 128 input channels (A/B timing).
    {
      static const int gen4 = getenv("XV_BF16_GEN4") != nullptr ? atoi(getenv("XV_BF16_GEN4")) : 1;
      // (the data-gradient epilogue -- addend + relu mask -- exists in the 16x16 forms: XV_DGRAD_GEN4=0 keeps those convs on
      // generation 2, A/B timing)
      static const bool dgrad4 = getenv("XV_DGRAD_GEN4") == nullptr || atoi(getenv("XV_DGRAD_GEN4")) != 0;
      const bool dg = a.mask != nullptr || a.addend != nullptr;
      // With a stream-K workspace (the `streamk` latency option) only launches that fill the chip at least twice take
      // generations 4 / 5 (no tail to split there: conv1_2 .. conv3_3 at one image); the others keep generation 2 and its
      // stream-K tail (conv4_x / conv5_x at one image: 72 / 24 tiles for 256 CUs).
      const int64_t items16 = (int64_t)a.N * ((a.H + 15) / 16) * ((a.W + 31) / 32) * (a.Cout / 64);
      const bool no_tail = a.sk_ws == nullptr || items16 >= 2 * (int64_t)a.num_cus;
      // (XV_COL_ROUNDS=1, A/B timing: maps that tile both ways take the 24x16 tile when it makes whole rounds of workgroups
      // and the 16x32 tile does not -- conv4_x at 16 images: 6 rounds against 4.5)
      static const bool col_rounds = getenv("XV_COL_ROUNDS") != nullptr && atoi(getenv("XV_COL_ROUNDS")) != 0;
      if (col_rounds && gen4 && (!dg || dgrad4) && no_tail && !a.in_f8 && !a.out_f8 && a.pooled == nullptr &&
          xv_conv3x3_col_ok(a.H, a.W, a.Cin, a.Cout, 3) && xv_conv3x3_dma4_exact(a.H, a.W)) {
        const int64_t i26 = (int64_t)a.N * (a.H / 16) * (a.W / 32) * (a.Cout / 64), i27 = (int64_t)a.N * (a.H / 24) * (a.W / 16) * (a.Cout / 64);
        if (i27 % a.num_cus == 0 && i26 % a.num_cus != 0) return 27;
      }
      if (gen4 && (!dg || (dgrad4 && a.pooled == nullptr)) && no_tail && !a.in_f8 && !a.out_f8 &&
          xv_conv3x3_dma4_bf16_ok(a.H, a.W, a.Cin, a.Cout) && xv_conv3x3_dma4_exact(a.H, a.W) && (gen4 != 2 || a.Cin >= 128)) {
        // Latency rule (round 5, batch 1): below two rounds of 16x32 items a launch ends when its last round does, and the
        // 24x16 tile of generation 5 (same loop, same bits) makes more and smaller items -- conv2_1 at one image: 288 items
        // = 2 rounds of 512 pixels against 384 items = 2 rounds of 384; conv3_x / conv4_x: one round either way.
        // tools/conv_tune.py --batch 1: conv2_1 694 against 571 TFLOP/s, conv3_2 974 / 840, conv4_2 578 / 486
        // (profiles/r5_conv_tune_b1.txt).  The 24x16 tile is ~5 % slower per pixel once the chip is full.  XV_CFG_LATENCY=0: off.
        static const bool latency_rule = getenv("XV_CFG_LATENCY") == nullptr || atoi(getenv("XV_CFG_LATENCY")) != 0;
        if (latency_rule && a.pooled == nullptr && a.sk_ws == nullptr && items16 < 2 * (int64_t)a.num_cus &&
            xv_conv3x3_col_ok(a.H, a.W, a.Cin, a.Cout, 3)) {
          const int64_t items24 = (int64_t)a.N * (a.H / 24) * (a.W / 16) * (a.Cout / 64);
          const double c26 = (double)((items16 + a.num_cus - 1) / a.num_cus) * 512.0;
          const double c27 = (double)((items24 + a.num_cus - 1) / a.num_cus) * 384.0 / 0.95;
          if (c27 < c26) return 27;
        }
        return 26;
      }
      // maps that tile in 24x16 but not in 16x32 (the 24x48 conv5 maps of a 768x384 input): the same loop on a column of
      // waves, generation 5 -- conv5_1 at 16 images 1 190 against 985 TFLOP/s on generation 2's 24x16 tile (configuration 22)
      if (gen4 && (!dg || dgrad4) && no_tail && !a.in_f8 && !a.out_f8 && a.pooled == nullptr &&
          xv_conv3x3_col_ok(a.H, a.W, a.Cin, a.Cout, 3))
        return 27;
    }
    int g2 = 17;
    double s2 = 1.25;
    if (a.pooled == nullptr && round_cost(22, 1.15) < round_cost(17, 1.25)) g2 = 22, s2 = 1.15;
    if (covered(g2) <= s2 * g1) return g2;
  }
  // 1x1 convs (plain GEMMs, AdapNet's block stages): 128 output channels per workgroup halve the activation re-reads
  // (tools/conv1x1_tune.py: 1.1-1.7x over the 64-channel tiles from 128 input channels up); from 256 input channels
  // the flat-GEMM kernel (generation 3) is ahead by another 1.1-1.5x
  if (k == 1 && a.Cout % 128 == 0 && a.Cin >= 256 && a.pooled == nullptr && a.y != nullptr) return 18;
  // ... and its narrow form for 64 output channels (the FCN's score convs: 20-28 us -> see conv1x1_gemm.hip)
  if (k == 1 && a.Cout == 64 && a.Cin >= 128 && a.pooled == nullptr && a.y != nullptr) return 23;
  if (k == 1 && a.Cout % 128 == 0 && a.Cin >= 128 && covered(1) <= 1.1 * g1) return 1;
  return covered(15) < covered(14) ? 15 : 14;
}

size_t streamk_workspace_bytes() { return (size_t)XV_SK_HDR + (size_t)2 * xv_num_cus() * XV_SK_SLAB; }

int conv_fwd_impl(const xv_act* x, const void* w_packed, const float* bias, const xv_act* y, const xv_act* pooled,
                  int k, int relu, int cfg, void* stream, const __bf16* mask = nullptr,
                  const __bf16* addend = nullptr, void* workspace = nullptr, size_t workspace_bytes = 0,
                  void* split_ws = nullptr, size_t split_bytes = 0) {
  XV_CHECK_ARG(x && x->data && w_packed && bias && y);
  XV_CHECK_ARG(y->data || (pooled && pooled->data));
  XV_CHECK_SHAPE(k == 1 || k == 3);
  XV_CHECK_SHAPE(xv_dims_sane(x->n, x->h, x->w) && x->c > 0 && (x->c & 63) == 0);
  XV_CHECK_SHAPE(y->n == x->n && y->h == x->h && y->w == x->w && y->c > 0 && (y->c & 63) == 0);
  XV_CHECK_ARG((x->dtype == XV_BF16 || x->dtype == XV_FP8) && (y->dtype == XV_BF16 || y->dtype == XV_FP8));
  const int in_f8 = x->dtype == XV_FP8, out_f8 = y->dtype == XV_FP8;
  // e4m3 operands: 128-channel chunks (first generation); a 3x3 conv on the generation-4 kernel takes 64-channel chunks
  if (in_f8) XV_CHECK_SHAPE((x->c & (k == 3 ? 63 : 127)) == 0 && x->scale_exp > -127 && x->scale_exp < 127);
  if (in_f8 || out_f8) XV_CHECK_SHAPE(mask == nullptr && addend == nullptr);  // forward only
  if (out_f8) XV_CHECK_SHAPE(y->scale_exp > -100 && y->scale_exp < 100);
  if (pooled && pooled->data) XV_CHECK_ARG(pooled->dtype == y->dtype && pooled->scale_exp == y->scale_exp);
  XV_CHECK_ARG((((uintptr_t)x->data | (uintptr_t)w_packed | (uintptr_t)bias | (uintptr_t)y->data) & 15) == 0);
  ConvArgs a{};
  a.x = (const __bf16*)x->data;
  a.wpk = (const __bf16*)w_packed;
  a.bias = bias;
  a.y = (__bf16*)y->data;
  a.pooled = nullptr;
  a.mask = mask;
  a.addend = addend;
  a.N = x->n;
  a.H = x->h;
  a.W = x->w;
  a.Cin = x->c;
  a.Cout = y->c;
  a.relu = relu;
  a.num_cus = xv_num_cus();
  a.in_f8 = in_f8;
  a.out_f8 = out_f8;
  a.scale_x = in_f8 ? ((127 + x->scale_exp) & 0xff) * 0x01010101 : 0;
  a.out_mul = out_f8 ? exp2f((float)-y->scale_exp) : 1.f;
  if (workspace != nullptr) {
    XV_CHECK_ARG(((uintptr_t)workspace & 15) == 0);
    if (workspace_bytes < streamk_workspace_bytes()) return XV_EWORKSPACE;
    a.sk_ws = (char*)workspace;
  }
  if (split_ws != nullptr) {
    XV_CHECK_ARG(((uintptr_t)split_ws & 15) == 0);
    a.split_ws = split_ws, a.split_bytes = split_bytes;
  }
  if (pooled && pooled->data) {
    XV_CHECK_SHAPE(k == 3 && (x->h & 1) == 0 && (x->w & 1) == 0);
    XV_CHECK_SHAPE(pooled->n == x->n && pooled->h == x->h / 2 && pooled->w == x->w / 2 && pooled->c == y->c);
    XV_CHECK_ARG(((uintptr_t)pooled->data & 15) == 0);
    a.pooled = (__bf16*)pooled->data;
  }
  if (cfg < 0) cfg = pick_cfg(a, k);
  if (cfg < 0) return XV_ESHAPE;
  if (in_f8 && (a.Cin & 127) && cfg != 24) return XV_ESHAPE;  // 64-channel e4m3 chunks exist in generation 4 only
  hipStream_t s = (hipStream_t)stream;
  return k == 3 ? launch_cfg<3>(cfg, a, s) : launch_cfg<1>(cfg, a, s);
}

}  // namespace

extern "C" size_t xv_packed_weight_bytes(int k, int cin, int cout) {
  if ((k != 1 && k != 3) || cin <= 0 || cout <= 0 || (cin & 63) || (cout & 63)) return 0;
  return (size_t)k * k * cin * cout * 2 * (k == 3 ? 3 : 1);  // 3x3: the three packed images (generations 1, 2, 4 / 5 on 16x16 blocks)
}

extern "C" size_t xv_packed_weight_bytes_f8(int k, int cin, int cout) {
  if ((k != 1 && k != 3) || cin <= 0 || cout <= 0 || (cin & (k == 3 ? 63 : 127)) || (cout & 63)) return 0;
  // 3x3: the generation-1 image (left empty when cin is not a multiple of 128) and the generation-4 image
  return 256 + (size_t)k * k * cin * cout * (k == 3 ? 2 : 1);
}

extern "C" int xv_pack_conv_weights_f8(const float* w_hwio, void* packed, int k, int cin, int cout, int scale_exp,
                                        void* stream) {
  XV_CHECK_ARG(w_hwio && packed && (((uintptr_t)packed) & 15) == 0);
  XV_CHECK_SHAPE((k == 1 || k == 3) && cin > 0 && cout > 0 && (cin & (k == 3 ? 63 : 127)) == 0 && (cout & 63) == 0);
  XV_CHECK_SHAPE(scale_exp > -127 && scale_exp < 127);
  const int64_t total4 = (int64_t)k * k * cin * cout / 4;
  const int blocks = (int)((total4 + 255) / 256 < 4096 ? (total4 + 255) / 256 : 4096);
  if ((cin & 127) == 0)
    hipLaunchKernelGGL(pack_weights_f8_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_hwio, (char*)packed, k * k,
                       cin, cout, scale_exp, exp2f((float)-scale_exp));
  else  // no 128-channel image: header only (scale exponent, zero padding)
    hipLaunchKernelGGL(pack_weights_f8_header_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (char*)packed, scale_exp);
  if (k == 3)
    xv_launch_pack_weights_f8_g4(w_hwio, (char*)packed + 256 + (size_t)9 * cin * cout, 9, cin, cout, exp2f((float)-scale_exp),
                                 (hipStream_t)stream);
  return xv_launch_status();
}

extern "C" int xv_pack_conv_weights(const float* w_hwio, void* packed, int k, int cin, int cout, void* stream) {
  XV_CHECK_ARG(w_hwio && packed);
  XV_CHECK_SHAPE((k == 1 || k == 3) && cin > 0 && cout > 0 && (cin & 63) == 0 && (cout & 63) == 0);
  const int64_t total = (int64_t)k * k * cin * cout;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_hwio, (__bf16*)packed,
                     k * k, cin, cout, 0, (__bf16*)nullptr);
  return xv_launch_status();
}

extern "C" int xv_pack_conv_weights_pair(const float* w_hwio, void* packed, void* packed_dgrad, int k, int cin, int cout,
                                          void* stream) {
  XV_CHECK_ARG(w_hwio && packed && packed_dgrad);
  XV_CHECK_SHAPE((k == 1 || k == 3) && cin > 0 && cout > 0 && (cin & 63) == 0 && (cout & 63) == 0);
  const int64_t total = (int64_t)k * k * cin * cout;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_hwio, (__bf16*)packed,
                     k * k, cin, cout, 0, (__bf16*)packed_dgrad);
  return xv_launch_status();
}

extern "C" int xv_pack_conv_weights_multi(const xv_pack_desc* table_device, int n, void* stream) {
  XV_CHECK_ARG(table_device != nullptr && n > 0 && n <= 65535);
  hipLaunchKernelGGL(pack_weights_multi_kernel, dim3(576, (unsigned)n), dim3(256), 0, (hipStream_t)stream, table_device);
  return xv_launch_status();
}

extern "C" int xv_memset_zero(void* p, size_t bytes, void* stream) {
  XV_CHECK_ARG(p != nullptr);
  const hipError_t e = hipMemsetAsync(p, 0, bytes, (hipStream_t)stream);
  return e == hipSuccess ? XV_OK : (int)e;
}

extern "C" int xv_pack_conv_weights_dgrad(const float* w_hwio, void* packed, int k, int cin, int cout, void* stream) {
  XV_CHECK_ARG(w_hwio && packed);
  XV_CHECK_SHAPE((k == 1 || k == 3) && cin > 0 && cout > 0 && (cin & 63) == 0 && (cout & 63) == 0);
  const int64_t total = (int64_t)k * k * cin * cout;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_hwio, (__bf16*)packed,
                     k * k, cin, cout, 1, (__bf16*)nullptr);
  return xv_launch_status();
}

// dx = (conv(dy, Wd) [+ addend]) masked by relu_ref > 0: Conv2DBackpropInput of tf.layers.conv2d followed by
// the ReluGrad of the layer below (and the AddN where two gradient paths meet), in the forward kernel.
extern "C" int xv_conv2d_bwd_data(const xv_act* dy, const void* w_packed_dgrad, const float* zero_bias,
                                  const xv_act* relu_ref, const xv_act* addend, const xv_act* dx, int k,
                                  void* stream) {
  XV_REQUIRE_BF16(dy, relu_ref, addend, dx);
  XV_CHECK_ARG(dx && dx->data && zero_bias);
  if (relu_ref && relu_ref->data)
    XV_CHECK_SHAPE(relu_ref->n == dx->n && relu_ref->h == dx->h && relu_ref->w == dx->w && relu_ref->c == dx->c);
  if (addend && addend->data)
    XV_CHECK_SHAPE(addend->n == dx->n && addend->h == dx->h && addend->w == dx->w && addend->c == dx->c);
  return conv_fwd_impl(dy, w_packed_dgrad, zero_bias, dx, nullptr, k, 0, -1, stream,
                       relu_ref ? (const __bf16*)relu_ref->data : nullptr,
                       addend ? (const __bf16*)addend->data : nullptr);
}

extern "C" int xv_conv2d_fwd_residual(const xv_act* x, const void* w_packed, const float* bias,
                                      const xv_act* residual, const xv_act* y, int relu, void* stream) {
  XV_REQUIRE_BF16(x, residual, y);
  // 1x1 only: those shapes always run on the first-generation kernel, whose epilogue applies the activation before
  // the addend (the second generation clamps at the store, after it)
  XV_CHECK_ARG(y && y->data && residual && residual->data);
  XV_CHECK_SHAPE(residual->n == y->n && residual->h == y->h && residual->w == y->w && residual->c == y->c);
  return conv_fwd_impl(x, w_packed, bias, y, nullptr, 1, relu, -1, stream, nullptr, (const __bf16*)residual->data);
}

// pointwise.hip
int xv_launch_depth_to_space(const xv_act* z, const float* scale, const float* shift, const xv_act* residual, const xv_act* y,
                             int stride, int relu, hipStream_t stream);

// The 3x3 conv of a conv -> batch norm block with the batch statistics taken in the conv's own epilogue (generation 4, bf16
// maps that tile exactly in 16x32, at most 512 output channels): y = conv(x, W) + b as xv_conv2d_fwd(relu = 0), and row
// w of `stats_rows` ([xv_conv2d_stats_rows()][2 cout] floats) = workgroup w's per-channel sum and sum of squares of the
// stored outputs; xv_bn_sums_from_rows adds the rows in a fixed tree.  XV_ESHAPE where that kernel does not apply: the
// caller then uses xv_conv2d_fwd + xv_bn_stats.
extern "C" int xv_conv2d_stats_rows(void) { return xv_num_cus(); }

extern "C" int xv_conv2d_fwd_stats(const xv_act* x, const void* w_packed, const float* bias, const xv_act* y, float* stats_rows,
                                   size_t stats_bytes, void* stream) {
  XV_REQUIRE_BF16(x, y);
  XV_CHECK_ARG(x && x->data && w_packed && bias && y && y->data && stats_rows);
  XV_CHECK_ARG((((uintptr_t)x->data | (uintptr_t)w_packed | (uintptr_t)bias | (uintptr_t)y->data | (uintptr_t)stats_rows) & 15) == 0);
  XV_CHECK_SHAPE(x->n > 0 && y->n == x->n && y->h == x->h && y->w == x->w && (x->c & 63) == 0 && (y->c & 63) == 0);
  if (!xv_conv3x3_dma4_bf16_ok(x->h, x->w, x->c, y->c) || y->c > 512) return XV_ESHAPE;
  if (stats_bytes < (size_t)xv_num_cus() * 2 * y->c * sizeof(float)) return XV_EWORKSPACE;
  return xv_launch_conv3x3_f8_dma(x->data, w_packed, bias, y->data, nullptr, x->n, x->h, x->w, x->c, y->c, 0, 0, 0, 0, 1.f,
                                  xv_num_cus(), (hipStream_t)stream, stats_rows, 1);
}

// THE ROUTED POOL OF TRAINING (generation 4, bf16 maps that tile exactly in 16x32 pixels; XV_ESHAPE elsewhere: the caller then
// keeps the full map and xv_maxpool2x2_bwd).  The full-resolution output of a conv in front of a pool is read by nothing
// but MaxPoolGrad (max_pooling2d under the relu of tf.layers.conv2d: simple_fcn.py:41,44,48): the forward conv writes the
// pooled map and one ROUTE BYTE per pooled value -- 0 where the window's maximum is not positive, else 0x80 >> the first
// position of the maximum in MaxPoolGrad's order -- and the data-gradient conv of the layer behind the pool stores its result
// THROUGH the routes onto the full-resolution gradient map (zeros at the three other positions): 0.5 + 4 bytes per pooled
// value of traffic where the full map (4 written, 4 read), the pooled gradient (1 + 1) and the routed map (4) made 14.
extern "C" size_t xv_conv2d_route_bytes(int n, int h, int w, int cout) {
  if (!xv_dims_sane(n, h, w) || cout <= 0 || (h & 1) || (w & 1)) return 0;
  return (size_t)n * (h / 2) * (w / 2) * cout;
}

// pooled = maxpool2x2(relu(conv3x3(x) + bias)), route = its route bytes [n][h/2][w/2][cout]
extern "C" int xv_conv2d_fwd_route(const xv_act* x, const void* w_packed, const float* bias, const xv_act* pooled, void* route,
                                   size_t route_bytes, void* stream) {
  XV_REQUIRE_BF16(x, pooled);
  XV_CHECK_ARG(x && x->data && w_packed && bias && pooled && pooled->data && route);
  XV_CHECK_ARG((((uintptr_t)x->data | (uintptr_t)w_packed | (uintptr_t)bias | (uintptr_t)pooled->data | (uintptr_t)route) & 15) == 0);
  XV_CHECK_SHAPE(x->n > 0 && x->h > 0 && x->w > 0 && (x->c & 63) == 0 && (pooled->c & 63) == 0 && pooled->c > 0);
  XV_CHECK_SHAPE(pooled->n == x->n && 2 * pooled->h == x->h && 2 * pooled->w == x->w);
  if (!xv_conv3x3_dma4_bf16_ok(x->h, x->w, x->c, pooled->c) || !xv_conv3x3_dma4_exact(x->h, x->w)) return XV_ESHAPE;
  if (route_bytes < xv_conv2d_route_bytes(x->n, x->h, x->w, pooled->c)) return XV_EWORKSPACE;
  return xv_launch_conv3x3_dma4_route(x->data, w_packed, bias, pooled->data, route, 0, x->n, x->h, x->w, x->c, pooled->c,
                                      xv_num_cus(), (hipStream_t)stream);
}

// dx (the map of twice dy's size) = MaxPoolGrad + ReluGrad of conv3x3(dy, Wd) through `route` ([n][dy.h][dy.w][dx.c] bytes of
// xv_conv2d_fwd_route): the same bits as xv_conv2d_bwd_data onto a pooled-size map followed by xv_maxpool2x2_bwd
extern "C" int xv_conv2d_bwd_data_route(const xv_act* dy, const void* w_packed_dgrad, const float* zero_bias, const void* route,
                                        size_t route_bytes, const xv_act* dx, void* stream) {
  XV_REQUIRE_BF16(dy, dx);
  XV_CHECK_ARG(dy && dy->data && w_packed_dgrad && zero_bias && route && dx && dx->data);
  XV_CHECK_ARG((((uintptr_t)dy->data | (uintptr_t)w_packed_dgrad | (uintptr_t)zero_bias | (uintptr_t)dx->data | (uintptr_t)route) & 15) == 0);
  XV_CHECK_SHAPE(dy->n > 0 && dy->h > 0 && dy->w > 0 && (dy->c & 63) == 0 && (dx->c & 63) == 0 && dx->c > 0);
  XV_CHECK_SHAPE(dx->n == dy->n && dx->h == 2 * dy->h && dx->w == 2 * dy->w);
  if (!xv_conv3x3_dma4_bf16_ok(dy->h, dy->w, dy->c, dx->c) || !xv_conv3x3_dma4_exact(dy->h, dy->w)) return XV_ESHAPE;
  if (route_bytes < xv_conv2d_route_bytes(dx->n, dx->h, dx->w, dx->c)) return XV_EWORKSPACE;
  return xv_launch_conv3x3_dma4_route(dy->data, w_packed_dgrad, zero_bias, dx->data, const_cast<void*>(route), 1, dy->n, dy->h,
                                      dy->w, dy->c, dx->c, xv_num_cus(), (hipStream_t)stream);
}

extern "C" size_t xv_deconv_dense_workspace_bytes(int n, int h, int w, int cout, int stride) {
  if (!xv_dims_sane(n, h, w) || cout <= 0 || stride <= 0 || stride > 64 || cout > (1 << 20)) return 0;
  return (size_t)n * ((size_t)h + 2) * ((size_t)w + 2) * stride * stride * cout * 2;
}

// A k x k / stride s transposed conv with k = 2s ('same': pad (k - s) / 2) is, per output phase (py, px), a 2x2-tap
// stride-1 conv of the input -- the taps of every phase fit the 3x3 window, so all s*s phases run as ONE 3x3 MFMA conv
// cin -> s*s*cout at the INPUT resolution (custom_layers.dense_deconv_as_conv3x3 arranges the kernel), followed by a
// depth-to-space pass that carries the batch norm, the activation and the residual add.
extern "C" int xv_deconv_dense_fwd(const xv_act* x, const void* w_phases_packed, const float* zero_bias, const float* scale,
                                   const float* shift, const xv_act* residual, const xv_act* y, int stride, int relu,
                                   void* workspace, size_t workspace_bytes, void* stream) {
  XV_REQUIRE_BF16(x, residual, y);
  XV_CHECK_ARG(x && x->data && y && y->data && w_phases_packed && zero_bias && workspace);
  XV_CHECK_ARG((scale == nullptr) == (shift == nullptr) && (((uintptr_t)workspace) & 15) == 0);
  XV_CHECK_SHAPE(stride >= 1 && stride <= 8 && y->n == x->n && y->h == x->h * stride && y->w == x->w * stride);
  XV_CHECK_SHAPE((y->c & 7) == 0 && ((stride * stride * y->c) & 63) == 0 && x->dtype == XV_BF16 && y->dtype == XV_BF16);
  if (residual && residual->data)
    XV_CHECK_SHAPE(residual->n == y->n && residual->h == y->h && residual->w == y->w && residual->c == y->c);
  if (workspace_bytes < xv_deconv_dense_workspace_bytes(x->n, x->h, x->w, y->c, stride)) return XV_EWORKSPACE;
  // the phase map: padded NHWC like every activation; its border is never read (depth-to-space reads the interior)
  xv_act z{workspace, x->n, x->h, x->w, stride * stride * y->c, XV_BF16, 0};
  const int rc = conv_fwd_impl(x, w_phases_packed, zero_bias, &z, nullptr, 3, 0, -1, stream);
  if (rc != XV_OK) return rc;
  return xv_launch_depth_to_space(&z, scale, shift, residual, y, stride, relu, (hipStream_t)stream);
}

extern "C" int xv_conv2d_fwd(const xv_act* x, const void* w_packed, const float* bias, const xv_act* y,
                             const xv_act* pooled, int k, int relu, void* stream) {
  return conv_fwd_impl(x, w_packed, bias, y, pooled, k, relu, -1, stream);
}

extern "C" int xv_conv2d_fwd_cfg(const xv_act* x, const void* w_packed, const float* bias, const xv_act* y,
                                 const xv_act* pooled, int k, int relu, int cfg, void* stream) {
  return conv_fwd_impl(x, w_packed, bias, y, pooled, k, relu, cfg, stream);
}

// The same 3x3 layer of TWO models (the two experts of a fusion model: own maps, weights and bias, one shape) in ONE launch
// of the persistent generation-4 / 5 kernel: the tile lists are concatenated, so the launch makes whole rounds of workgroups
// where each model alone leaves its last round half empty (conv4_x at 16 images of 768x384: 2 x 1152 tiles = 9 rounds of 256
// against 2 x 5; conv5_x: 3 against 2 x 2).  Results are bit-identical to two xv_conv2d_fwd calls.  XV_ESHAPE where the shape
// does not take configuration 26 / 27 / 28 (the caller then launches the two convs separately).
extern "C" int xv_conv2d_fwd_pair(const xv_act* xa, const void* wa_packed, const float* bias_a, const xv_act* ya,
                                  const xv_act* pooled_a, const xv_act* xb, const void* wb_packed, const float* bias_b,
                                  const xv_act* yb, const xv_act* pooled_b, int relu, void* stream) {
  XV_CHECK_ARG(xa && xa->data && xb && xb->data && wa_packed && wb_packed && bias_a && bias_b && ya && yb);
  const bool pool = pooled_a && pooled_a->data;
  XV_CHECK_ARG(pool == (pooled_b && pooled_b->data) && (ya->data != nullptr) == (yb->data != nullptr) && (ya->data || pool));
  XV_CHECK_SHAPE(xv_dims_sane(xa->n, xa->h, xa->w) && xa->c > 0 && (xa->c & 63) == 0 && ya->c > 0 && (ya->c & 63) == 0);
  XV_CHECK_SHAPE(xb->n == xa->n && xb->h == xa->h && xb->w == xa->w && xb->c == xa->c);
  XV_CHECK_SHAPE(ya->n == xa->n && ya->h == xa->h && ya->w == xa->w && yb->n == xa->n && yb->h == xa->h && yb->w == xa->w &&
                 yb->c == ya->c);
  XV_CHECK_ARG(xa->dtype == XV_BF16 && xb->dtype == XV_BF16 && ya->dtype == XV_BF16 && yb->dtype == XV_BF16);
  XV_CHECK_ARG((((uintptr_t)xa->data | (uintptr_t)xb->data | (uintptr_t)wa_packed | (uintptr_t)wb_packed | (uintptr_t)bias_a |
                 (uintptr_t)bias_b | (uintptr_t)ya->data | (uintptr_t)yb->data) & 15) == 0);
  if (pool) {
    XV_CHECK_SHAPE((xa->h & 1) == 0 && (xa->w & 1) == 0);
    XV_CHECK_SHAPE(pooled_a->n == xa->n && pooled_a->h == xa->h / 2 && pooled_a->w == xa->w / 2 && pooled_a->c == ya->c &&
                   pooled_b->n == xa->n && pooled_b->h == xa->h / 2 && pooled_b->w == xa->w / 2 && pooled_b->c == ya->c);
    XV_CHECK_ARG(pooled_a->dtype == XV_BF16 && pooled_b->dtype == XV_BF16 &&
                 (((uintptr_t)pooled_a->data | (uintptr_t)pooled_b->data) & 15) == 0);
  }
  ConvArgs a{};
  a.x = (const __bf16*)xa->data, a.y = (__bf16*)ya->data, a.pooled = pool ? (__bf16*)pooled_a->data : nullptr;
  a.N = xa->n, a.H = xa->h, a.W = xa->w, a.Cin = xa->c, a.Cout = ya->c;
  a.relu = relu;
  a.num_cus = xv_num_cus();
  const int cfg = pick_cfg(a, 3);
  const void* const x[2] = {xa->data, xb->data};
  const void* const w[2] = {wa_packed, wb_packed};
  const float* const b[2] = {bias_a, bias_b};
  void* const y[2] = {ya->data, yb->data};
  void* const q[2] = {pool ? pooled_a->data : nullptr, pool ? pooled_b->data : nullptr};
  if (cfg == 26)
    return xv_launch_conv3x3_dma4_pair(x, w, b, y, q, a.N, a.H, a.W, a.Cin, a.Cout, relu, a.num_cus, (hipStream_t)stream);
  if (cfg == 27 || cfg == 28)
    return xv_launch_conv3x3_col_pair(x, w, b, y, q, a.N, a.H, a.W, a.Cin, a.Cout, relu, cfg == 27 ? 3 : 4, a.num_cus,
                                      (hipStream_t)stream);
  return XV_ESHAPE;
}

extern "C" size_t xv_conv2d_streamk_workspace_bytes(void) { return streamk_workspace_bytes(); }

extern "C" int xv_conv2d_fwd_ws(const xv_act* x, const void* w_packed, const float* bias, const xv_act* y,
                                const xv_act* pooled, int k, int relu, int cfg, void* workspace, size_t workspace_bytes,
                                void* stream) {
  return conv_fwd_impl(x, w_packed, bias, y, pooled, k, relu, cfg, stream, nullptr, nullptr, workspace, workspace_bytes);
}

// Forward conv with a workspace for the SPLIT form of generation 5 (conv_col_dma.hip: layers whose whole tiles fill less than
// half the CUs -- conv5_x of a 768x384 input at one or two images -- run one work item per (tile, chunk group) and add the
// groups' fp32 slabs in a second launch; the same bits as the unsplit launch of the same layer at any batch size).
extern "C" size_t xv_conv2d_split_workspace_bytes(int n, int h, int w, int cin, int cout) {
  if (!xv_dims_sane(n, h, w) || cin <= 0 || cout <= 0) return 0;
  return xv_conv3x3_col_split_bytes(n, h, w, cin, cout, 3, xv_num_cus());
}

extern "C" int xv_conv2d_fwd_split(const xv_act* x, const void* w_packed, const float* bias, const xv_act* y,
                                   const xv_act* pooled, int k, int relu, int cfg, void* split_workspace,
                                   size_t split_workspace_bytes, void* stream) {
  return conv_fwd_impl(x, w_packed, bias, y, pooled, k, relu, cfg, stream, nullptr, nullptr, nullptr, 0, split_workspace,
                       split_workspace_bytes);
}

extern "C" int xv_conv2d_bwd_data_ws(const xv_act* dy, const void* w_packed_dgrad, const float* zero_bias,
                                     const xv_act* relu_ref, const xv_act* addend, const xv_act* dx, int k, void* workspace,
                                     size_t workspace_bytes, void* stream) {
  XV_REQUIRE_BF16(dy, relu_ref, addend, dx);
  XV_CHECK_ARG(dx && dx->data && zero_bias);
  if (relu_ref && relu_ref->data)
    XV_CHECK_SHAPE(relu_ref->n == dx->n && relu_ref->h == dx->h && relu_ref->w == dx->w && relu_ref->c == dx->c);
  if (addend && addend->data)
    XV_CHECK_SHAPE(addend->n == dx->n && addend->h == dx->h && addend->w == dx->w && addend->c == dx->c);
  return conv_fwd_impl(dy, w_packed_dgrad, zero_bias, dx, nullptr, k, 0, -1, stream,
                       relu_ref ? (const __bf16*)relu_ref->data : nullptr,
                       addend ? (const __bf16*)addend->data : nullptr, workspace, workspace_bytes);
}

extern "C" int xv_conv2d_num_cfgs(void) { return XV_NUM_CONV_CFG; }

// The tile configuration xv_conv2d_fwd (cfg = -1) / xv_conv2d_bwd_data would choose for this shape, without launching
// anything (host arithmetic only: the chooser is part of the contract the tests and the sanitizer build check).
// flags: bit 0 = with a pooled output, bit 1 = data-gradient epilogue (addend / relu mask), bit 2 = with a stream-K workspace.
extern "C" int xv_conv2d_choose_cfg(int n, int h, int w, int cin, int cout, int k, int in_dtype, int out_dtype, int flags) {
  XV_CHECK_SHAPE(k == 1 || k == 3);
  XV_CHECK_SHAPE(xv_dims_sane(n, h, w) && cin > 0 && cout > 0 && (cin & 63) == 0 && (cout & 63) == 0);
  XV_CHECK_ARG((in_dtype == XV_BF16 || in_dtype == XV_FP8) && (out_dtype == XV_BF16 || out_dtype == XV_FP8));
  static const __bf16 sentinel[8] = {};
  static char sk_sentinel[16];
  ConvArgs a{};
  a.N = n, a.H = h, a.W = w, a.Cin = cin, a.Cout = cout;
  a.relu = 1;
  a.num_cus = xv_num_cus();
  a.in_f8 = in_dtype == XV_FP8, a.out_f8 = out_dtype == XV_FP8;
  if (a.in_f8) XV_CHECK_SHAPE((cin & (k == 3 ? 63 : 127)) == 0);
  if (flags & 1) {
    XV_CHECK_SHAPE(k == 3 && (h & 1) == 0 && (w & 1) == 0);
    a.pooled = const_cast<__bf16*>(sentinel);
  }
  if (flags & 2) {
    XV_CHECK_SHAPE(!a.in_f8 && !a.out_f8);
    a.mask = sentinel, a.addend = sentinel;
  }
  if (flags & 4) a.sk_ws = sk_sentinel;
  const int cfg = pick_cfg(a, k);
  return cfg < 0 ? XV_ESHAPE : cfg;
}

#ifdef XV_CONV_TRACE
extern "C" int xv_debug_read_trace(void* dst, size_t bytes) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(xv_trace_buf), bytes);
}
#endif

#ifdef XV_CLOCK_STAMP
// [workgroup][s_memtime before, s_memrealtime before, s_memtime after, s_memrealtime after] of the last generation-2 launch
extern "C" int xv_debug_read_clock_g2(void* dst, size_t bytes) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(xv_clk_g2), bytes); }
extern "C" int xv_debug_reset_clock_g2(void) {
  static unsigned long long zeros[4 * XV_CLK_SLOTS];
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(xv_clk_g2), zeros, sizeof(zeros));
}
#endif
