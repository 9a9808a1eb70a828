// The FCN trunk in plain fp32 ("exact" mode, conv_dtype='fp32'): the reference graph's own arithmetic type
// (tf.layers.conv2d / max_pooling2d / conv2d_transpose on float32, simple_fcn.py:39-87, custom_layers.py:71-139) on dense
// unpadded NHWC float32 maps, no bf16 storage anywhere.  On the same weights its label maps must equal the fp32 oracle's,
// which turns "the bf16 path agrees with the oracle on clear margins" into evidence that the pixels the bf16 path loses are
// lost to bf16 storage and not to a kernel (tests/test_exact_f32_gpu.py).
//
//   xv_conv2d_f32        3x3 'same' / 1x1, bias, optional relu -- on the matrix cores: v_mfma_f32_32x32x2_f32, whose result
//   xv_conv2d_f32_pool   is bit for bit a k-ordered fp32 fmaf chain (MI355X_MICROARCH.md, Matrix cores: exact f32 at the f32
//                        vector rate, 157 TFLOP/s); _pool also writes the 2x2 max-pooled map from the accumulators
//   xv_conv2d_f32_scalar the round-4 kernel (fp32 FMAs on the vector ALU), kept as the A/B baseline of the bench record
//   xv_upsample2x_f32    y = residual + relu(bilinear_x2(x))  (upscore_conv5 + add_score, simple_fcn.py:82-85)
//   xv_score_lowres_f32  S = fused . Ws at 1/8 resolution into the padded [N][h+2][w+2][CP] layout the decoder-head
//                        kernels read (xv_decoder_head_fwd's first half, from an fp32 map)
#include "xv_common.h"

namespace {

// One workgroup = 64 output pixels (an 8x8 tile) x 64 output channels: wave w owns channels [16w, 16w+16), lane = pixel.
// Input channels go by chunks of 16: the chunk's weights [taps][16][64] sit in LDS (every lane of a wave reads the same
// 16 floats of a (tap, cin) row: broadcast reads), the pixel's own taps come straight from global memory (L1 / L2).
// Accumulation order per output: cin chunk outermost, then tap, then cin -- a fixed order, in fp32 FMAs.
template <int KS>
__global__ __launch_bounds__(256) void conv_f32_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ bias, float* __restrict__ y, int N, int H,
                                                      int W, int Cin, int Cout, int relu) {
  constexpr int TAPS = KS * KS, PAD = KS / 2;
  __shared__ __attribute__((aligned(16))) float ws[TAPS * 16 * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tiles_x = (W + 7) / 8, tiles_y = (H + 7) / 8;
  int t = blockIdx.x;
  const int tx = t % tiles_x;
  t /= tiles_x;
  const int ty = t % tiles_y;
  const int n = t / tiles_y;
  const int co0 = blockIdx.y * 64;
  const int py = ty * 8 + (lane >> 3), px = tx * 8 + (lane & 7);
  const bool live = py < H && px < W;
  float acc[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) acc[j] = 0.f;
  for (int c0 = 0; c0 < Cin; c0 += 16) {
    const int cc = Cin - c0 < 16 ? Cin - c0 : 16;
    __syncthreads();
    for (int i = threadIdx.x; i < TAPS * 16 * 64; i += 256) {
      const int co = i & 63, ci = (i >> 6) & 15, tap = i >> 10;
      ws[i] = (ci < cc && co0 + co < Cout) ? w[((int64_t)tap * Cin + c0 + ci) * Cout + co0 + co] : 0.f;
    }
    __syncthreads();
    if (!live) continue;
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
      const int yy = py + tap / KS - PAD, xx = px + tap % KS - PAD;
      if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;  // zero padding ('same')
      const float* src = x + (((int64_t)n * H + yy) * W + xx) * Cin + c0;
      for (int ci = 0; ci < cc; ++ci) {
        const float xv = src[ci];
        const float* wr = ws + (tap * 16 + ci) * 64 + wave * 16;
#pragma unroll
        for (int j4 = 0; j4 < 4; ++j4) {
          const f32x4 wv = *reinterpret_cast<const f32x4*>(wr + j4 * 4);
          acc[j4 * 4] = fmaf(xv, wv.x, acc[j4 * 4]);
          acc[j4 * 4 + 1] = fmaf(xv, wv.y, acc[j4 * 4 + 1]);
          acc[j4 * 4 + 2] = fmaf(xv, wv.z, acc[j4 * 4 + 2]);
          acc[j4 * 4 + 3] = fmaf(xv, wv.w, acc[j4 * 4 + 3]);
        }
      }
    }
  }
  if (!live) return;
  float* dst = y + (((int64_t)n * H + py) * W + px) * Cout + co0 + wave * 16;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    if (co0 + wave * 16 + j >= Cout) break;
    float v = acc[j] + bias[co0 + wave * 16 + j];
    if (relu) v = fmaxf(v, 0.f);
    dst[j] = v;
  }
}

// ---- the fp32 conv on the matrix cores ---------------------------------------------------------------------------------------
// Implicit GEMM D[cout][pixel] = sum_{tap,cin} W[tap][cin][cout] * X[pixel + tap][cin], weights as the MFMA A operand, pixels
// as B.  One 4-wave workgroup (one wave per SIMD) owns an 8 x 32-pixel x 64-channel tile; wave v owns tile rows 2v, 2v+1 and
// both 32-channel blocks: four 32x32 accumulators.  Input channels go by chunks of CK = 16: the chunk's halo patch
// [10][34][16] and its nine [64][16] weight tiles sit in LDS with a pixel / channel-row pitch of 20 dwords (80 bytes:
// 16-byte aligned, and 16 consecutive rows start in 16 different 4-bank groups, so every ds_read_b128 and the weight
// ds_write_b128s are conflict-free), double-buffered: the next chunk is requested into registers before the current chunk's
// first taps and stored into the other buffer after them -- one barrier per chunk (18 432 matrix-pipe cycles); the fragments
// of tap t + 1 are read while tap t multiplies.  The grid is persistent: a workgroup walks tiles b, b + grid, ... and the
// chunk pipeline runs across the tile boundary (the next tile's first chunk lands during the current tile's last one, the
// tile's stores drain under the next tile's taps).
// Maps with at most four input channels (conv1_1: raw RGB / depth) take chunks of CK = 4 at a pitch of 6 dwords
// (ds_read_b64: 32 consecutive pixels cover the 64 banks once): two k steps per tap instead of eight.
// k order inside a chunk: a lane half h (= lane >> 5, the instruction's k index) holds CK / 2 channels of its row in one
// fragment -- CK = 16: channels 8j + 4h + e (j = 0, 1: two 16-byte reads), step (j, e) multiplies channel 8j + e (h = 0) and
// 8j + 4 + e (h = 1); CK = 4: channels 2h + e -- the same map on both operands.
// Per output: chunk outermost, then tap, then the steps, each step adding its h = 0 product before its h = 1 product
// -- a fixed fp32 fmaf chain, independent of the batch, the tile and the launch geometry.
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KS, int CK>
struct F32Geo {
  static constexpr int TAPS = KS * KS, PAD = KS / 2, TH = 8, TW = 32, PH = TH + 2 * PAD, PW = TW + 2 * PAD;
  static constexpr int NQ = CK / 4, PS = CK == 16 ? 20 : 6, S = CK / 2, HOFF = CK == 16 ? 4 : 2;
  static constexpr int PATCH_DW = PH * PW * PS, WT_DW = TAPS * 64 * PS, BUF_DW = PATCH_DW + WT_DW;
  static constexpr int NPI = (PH * PW * NQ + 255) / 256;    // patch pieces (4 channels of a pixel) per thread and chunk
  static constexpr int NWI = (TAPS * NQ * 64 + 255) / 256;  // weight pieces (4 input channels of one output channel and tap)
  static constexpr int LDS_BYTES = 2 * BUF_DW * 4;
  static_assert(CK == 16 || CK == 4, "chunk sizes with a conflict-free pitch");
  static_assert(PATCH_DW % 4 == 0 && BUF_DW % 4 == 0, "16-byte aligned buffers");
};

__device__ static __forceinline__ void xv_lds_barrier() {
  // LDS-only rendezvous: __syncthreads() also waits for vmcnt(0), i.e. for the tile's global stores and the prefetched loads
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int KS, int CK, bool VEC>
__global__ __launch_bounds__(256) void conv_f32_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ y,
                                                           float* __restrict__ pooled, int H, int W, int Cin, int Cout, int relu,
                                                           int tiles_x, int tiles_y, int ncb, int total) {
  using G = F32Geo<KS, CK>;
  constexpr int TAPS = G::TAPS, PAD = G::PAD, TH = G::TH, TW = G::TW, PH = G::PH, PW = G::PW, PS = G::PS, NPI = G::NPI,
                NWI = G::NWI, NQ = G::NQ, S = G::S;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 31, hh = lane >> 5;

  struct Tile {
    int n, y0, x0, co0;
  };
  auto decode = [&](int t) {
    Tile r;
    const int cb = t % ncb;
    t /= ncb;
    const int tx = t % tiles_x;
    t /= tiles_x;
    r.y0 = __builtin_amdgcn_readfirstlane((t % tiles_y) * TH);
    r.n = __builtin_amdgcn_readfirstlane(t / tiles_y);
    r.x0 = __builtin_amdgcn_readfirstlane(tx * TW);
    r.co0 = __builtin_amdgcn_readfirstlane(cb * 64);
    return r;
  };

  // ---- staging: patch piece i of this thread = (pixel, channel quad), weight piece i = (tap, quad) of output channel w_co.
  // Every request is a buffer load through a bounds-checked descriptor (the image / the weight tensor): a masked piece --
  // halo pixels outside the map, channels past Cin, output channels past Cout -- gets an offset past the end and the hardware
  // returns zeros.  No conditional load (each compiles into a branch with its own s_waitcnt vmcnt(0), which serialises the
  // chunk's requests) and no value mask.  Offsets are 32-bit byte counts (sizes checked by the host).
  constexpr uint32_t OOB = 0x80000000u;
  uint32_t p_goff[NPI];            // byte offset of the piece's first channel inside its image, OOB: outside the map
  const int w_co = tid & 63;
  uint32_t s_wcol = OOB;           // byte offset of this thread's output channel inside a weight row, OOB: past Cout
  // descriptors from values the compiler can see are wave-uniform (readfirstlane of the base address halves and the size):
  // otherwise every buffer load is wrapped in a waterfall loop (cdna_hip_programming.md T20)
  auto uniform_rsrc = [](const float* base, int bytes) {
    const uint64_t a = (uint64_t)base;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(bytes),
                                             0x00020000);
  };
  auto img_rsrc = [&](int n) { return uniform_rsrc(x + (int64_t)n * H * W * Cin, H * W * Cin * 4); };
  auto s_img = img_rsrc(0);
  const auto w_rsrc = uniform_rsrc(w, TAPS * Cin * Cout * 4);
  auto stage_tile = [&](const Tile& t) {
#pragma unroll
    for (int i = 0; i < NPI; ++i) {
      const int item = tid + 256 * i, p = item / NQ, q = item % NQ;
      const int pr = p / PW, pc = p - pr * PW;
      const int gy = t.y0 + pr - PAD, gx = t.x0 + pc - PAD;
      const bool ok = (256 * (i + 1) <= PH * PW * NQ || item < PH * PW * NQ) && gy >= 0 && gy < H && gx >= 0 && gx < W;
      p_goff[i] = ok ? (uint32_t)((gy * W + gx) * Cin + 4 * q) * 4u : OOB;
    }
    s_img = img_rsrc(t.n);
    s_wcol = t.co0 + w_co < Cout ? (uint32_t)(t.co0 + w_co) * 4u : OOB;
  };
  // One staging piece = 4 channels of a patch pixel (pieces 0 .. NPI-1) or 4 input channels of one (tap, output channel)
  // (pieces NPI .. NP-1): requested into registers by load_piece, written to the other LDS buffer by store_piece.  The chunk
  // loop spreads the pieces over the taps so that the matrix pipe never waits for a block of staging instructions.
  constexpr int NP = NPI + NWI;
  f32x4 st_reg[NP];
  const uint32_t w_row = (uint32_t)Cout * 4u, w_tap = (uint32_t)Cin * w_row;
  auto ld1 = [](decltype(w_rsrc) r, uint32_t off) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0)); };
  auto load_piece = [&](int k, int c0) {
    if (k < NPI) {
      const int cbase = c0 + 4 * ((tid + 256 * k) % NQ);
      const uint32_t off = p_goff[k] + (uint32_t)c0 * 4u;        // OOB + a chunk offset stays past the end
      if (VEC) {
        st_reg[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(s_img, cbase < Cin ? off : OOB, 0, 0));
      } else {
        st_reg[k] = f32x4{ld1(s_img, cbase < Cin ? off : OOB), ld1(s_img, cbase + 1 < Cin ? off + 4 : OOB),
                          ld1(s_img, cbase + 2 < Cin ? off + 8 : OOB), ld1(s_img, cbase + 3 < Cin ? off + 12 : OOB)};
      }
    } else {
      const int r = wave + 4 * (k - NPI), ciq = r % NQ, tap = r / NQ;     // CK = 16: ciq = wave, tap = k - NPI
      const int ci = c0 + 4 * ciq;
      const uint32_t base = (uint32_t)tap * w_tap + (uint32_t)ci * w_row + s_wcol;
      const bool tap_ok = tap < TAPS;
      st_reg[k] = f32x4{ld1(w_rsrc, tap_ok && ci < Cin ? base : OOB), ld1(w_rsrc, tap_ok && ci + 1 < Cin ? base + w_row : OOB),
                        ld1(w_rsrc, tap_ok && ci + 2 < Cin ? base + 2 * w_row : OOB),
                        ld1(w_rsrc, tap_ok && ci + 3 < Cin ? base + 3 * w_row : OOB)};
    }
  };
  auto store4 = [](float* dst, const f32x4 v) {
    if (CK == 16) {
      *reinterpret_cast<f32x4*>(dst) = v;
    } else {        // 24-byte pitch: 8-byte aligned
      *reinterpret_cast<f32x2*>(dst) = f32x2{v.x, v.y};
      *reinterpret_cast<f32x2*>(dst + 2) = f32x2{v.z, v.w};
    }
  };
  auto store_piece = [&](int k, float* buf) {
    if (k < NPI) {
      const int item = tid + 256 * k;
      // (only the last piece index can fall past the patch: a compile-time test for the others keeps their stores branch-free)
      if (256 * (k + 1) <= PH * PW * NQ || item < PH * PW * NQ) store4(buf + (item / NQ) * PS + 4 * (item % NQ), st_reg[k]);
    } else {
      const int r = wave + 4 * (k - NPI), ciq = r % NQ, tap = r / NQ;
      if (tap < TAPS) store4(buf + G::PATCH_DW + (tap * 64 + w_co) * PS + 4 * ciq, st_reg[k]);
    }
  };

  // ---- fragments and the multiply ----------------------------------------------------------------------------------------
  struct Frag {
    float a[2][S], b[2][S];
  };
  const int r0 = 2 * wave;
  const int b_base = (r0 * PW + col) * PS + G::HOFF * hh;     // + ((j + ky) * PW + kx) * PS
  const int a_base = col * PS + G::HOFF * hh;                  // + (tap * 64 + 32 * blk) * PS
  auto read_frag = [&](const float* src, float (&dst)[S]) {
    if (CK == 16) {
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(src), v1 = *reinterpret_cast<const f32x4*>(src + 8);
      dst[0] = v0.x, dst[1] = v0.y, dst[2] = v0.z, dst[3] = v0.w;
      dst[S - 4] = v1.x, dst[S - 3] = v1.y, dst[S - 2] = v1.z, dst[S - 1] = v1.w;
    } else {
      const f32x2 v = *reinterpret_cast<const f32x2*>(src);
      dst[0] = v.x, dst[S - 1] = v.y;
    }
  };
  auto load_frags = [&](const float* buf, int tap, Frag& f) {
    const int ky = tap / KS, kx = tap % KS;
#pragma unroll
    for (int j = 0; j < 2; ++j) read_frag(buf + b_base + ((j + ky) * PW + kx) * PS, f.b[j]);
#pragma unroll
    for (int b = 0; b < 2; ++b) read_frag(buf + G::PATCH_DW + a_base + (tap * 64 + 32 * b) * PS, f.a[b]);
  };
  f32x16 acc[2][2];
  auto multiply = [&](const Frag& f) {
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[b][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[b][s], f.b[j][s], acc[b][j], 0, 0, 0);
          // the slice's other instructions (fragment reads, staging requests / stores and their address arithmetic), a few
          // behind every MFMA: each 64-cycle MFMA hides them, a block of them in front of the slice would not be hidden
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
          __builtin_amdgcn_sched_group_barrier(0x006, 3, 0);   // VALU | SALU
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read
          __builtin_amdgcn_sched_group_barrier(0x080, 1, 0);   // DS
        }
  };

  // ---- epilogue: register r of accumulator (b, j) = channel 32 b + 8 (r >> 2) + 4 hh + (r & 3) of pixel (r0 + j, col) -------
  const bool vec_out = (Cout & 3) == 0;
  auto store_out = [&](float* dst, const f32x4 v, int co) {
    if (vec_out) {
      *reinterpret_cast<f32x4*>(dst) = v;
    } else {
      dst[0] = v.x;
      if (co + 1 < Cout) dst[1] = v.y;
      if (co + 2 < Cout) dst[2] = v.z;
      if (co + 3 < Cout) dst[3] = v.w;
    }
  };
  auto epilogue = [&](const Tile& t) {
    const int px = t.x0 + col;
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int co = t.co0 + 32 * b + 8 * g + 4 * hh;
        if (co >= Cout) continue;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (vec_out) {
          bv = *reinterpret_cast<const f32x4*>(bias + co);
        } else {
          bv.x = bias[co];
          if (co + 1 < Cout) bv.y = bias[co + 1];
          if (co + 2 < Cout) bv.z = bias[co + 2];
          if (co + 3 < Cout) bv.w = bias[co + 3];
        }
        f32x4 v[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          v[j] = f32x4{acc[b][j][4 * g] + bv.x, acc[b][j][4 * g + 1] + bv.y, acc[b][j][4 * g + 2] + bv.z,
                       acc[b][j][4 * g + 3] + bv.w};
          if (relu) v[j] = f32x4{fmaxf(v[j].x, 0.f), fmaxf(v[j].y, 0.f), fmaxf(v[j].z, 0.f), fmaxf(v[j].w, 0.f)};
          const int py = t.y0 + r0 + j;
          if (y != nullptr && px < W && py < H) store_out(y + (((int64_t)t.n * H + py) * W + px) * Cout + co, v[j], co);
        }
        if (pooled != nullptr) {
          // 2x2 max: the two rows of this wave share the lane, the column neighbour is lane ^ 1 (H, W even: checked by the host)
          f32x4 m = {fmaxf(v[0].x, v[1].x), fmaxf(v[0].y, v[1].y), fmaxf(v[0].z, v[1].z), fmaxf(v[0].w, v[1].w)};
          m = f32x4{fmaxf(m.x, __shfl_xor(m.x, 1, 64)), fmaxf(m.y, __shfl_xor(m.y, 1, 64)), fmaxf(m.z, __shfl_xor(m.z, 1, 64)),
                    fmaxf(m.w, __shfl_xor(m.w, 1, 64))};
          const int py = t.y0 + r0;
          if ((col & 1) == 0 && px < W && py < H)
            store_out(pooled + (((int64_t)t.n * (H / 2) + py / 2) * (W / 2) + px / 2) * Cout + co, m, co);
        }
      }
  };

  // ---- the pipeline over (tile, chunk) -----------------------------------------------------------------------------------
  // A chunk's body is one basic block: the next chunk's requests (first half of the taps) and its LDS stores (second half)
  // are unconditional -- behind the last chunk of the last tile every offset is OOB and the stores land in the idle buffer.
  int t = blockIdx.x;
  if (t >= total) return;
  const int nchunks = (Cin + CK - 1) / CK;
  constexpr int HALF = TAPS / 2, LT = HALF > 0 ? HALF : 1, ST = TAPS - HALF;
  Tile cur = decode(t);
  stage_tile(cur);
#pragma unroll
  for (int k = 0; k < NP; ++k) load_piece(k, 0);
#pragma unroll
  for (int k = 0; k < NP; ++k) store_piece(k, lds);
  xv_lds_barrier();
  int it = 0;
  for (; t < total; t += gridDim.x) {
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][j][r] = 0.f;
    const int tn = t + (int)gridDim.x;
    const bool has_next = tn < total;
    for (int c = 0; c < nchunks; ++c, ++it) {
      const float* cbuf = lds + (it & 1) * G::BUF_DW;
      float* nbuf = lds + ((it + 1) & 1) * G::BUF_DW;
      const bool last = c + 1 == nchunks;
      if (last) {                                  // the next chunk is the next tile's first (or nothing)
        if (has_next) {
          stage_tile(decode(tn));
        } else {
#pragma unroll
          for (int i = 0; i < NPI; ++i) p_goff[i] = OOB;
          s_wcol = OOB;
        }
      }
      const int nc0 = last ? 0 : CK * (c + 1);
      Frag fr[2];
      load_frags(cbuf, 0, fr[0]);
      if (TAPS == 1) {
#pragma unroll
        for (int k = 0; k < NP; ++k) load_piece(k, nc0);
        multiply(fr[0]);
#pragma unroll
        for (int k = 0; k < NP; ++k) store_piece(k, nbuf);
      } else {
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
          if (tap + 1 < TAPS) load_frags(cbuf, tap + 1, fr[(tap + 1) & 1]);
#pragma unroll
          for (int k = 0; k < NP; ++k) {
            if (tap < HALF && k * LT / NP == tap) load_piece(k, nc0);
            if (tap >= HALF && k * ST / NP == tap - HALF) store_piece(k, nbuf);
          }
          multiply(fr[tap & 1]);
        }
      }
      xv_lds_barrier();
    }
    epilogue(cur);
    if (has_next) cur = decode(tn);
  }
}

__global__ __launch_bounds__(256) void maxpool_f32_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int Ho,
                                                         int Wo, int C) {
  const int64_t total = (int64_t)N * Ho * Wo * C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    int64_t r = i / C;
    const int ox = (int)(r % Wo);
    r /= Wo;
    const int oy = (int)(r % Ho);
    const int n = (int)(r / Ho);
    const float* p = x + (((int64_t)n * 2 * Ho + 2 * oy) * 2 * Wo + 2 * ox) * C + c;
    const int64_t row = (int64_t)2 * Wo * C;
    y[i] = fmaxf(fmaxf(p[0], p[C]), fmaxf(p[row], p[row + C]));
  }
}

// bilinear x2 transposed conv (k 4, stride 2, [TF1] 'same'): out[o] = sum_i in[i] * w1[o + 1 - 2 i], taps
// w1 = {.25, .75, .75, .25} (custom_layers.py:8-25); accumulated in the order of increasing source index like
// F.conv_transpose2d's scatter does not define -- any order of the (at most 4) products is within 1 ulp of another
// scale / shift (may be null): the inference batch norm between the deconv and its relu (custom_layers.py:112-119)
__global__ __launch_bounds__(256) void upsample2x_f32_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                            float* __restrict__ y, int N, int Hi, int Wi, int C,
                                                            const float* __restrict__ scale, const float* __restrict__ shift) {
  const int Ho = 2 * Hi, Wo = 2 * Wi;
  const int64_t total = (int64_t)N * Ho * Wo * C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    int64_t r = i / C;
    const int ox = (int)(r % Wo);
    r /= Wo;
    const int oy = (int)(r % Ho);
    const int n = (int)(r / Ho);
    // sources iy with 0 <= oy + 1 - 2 iy < 4: iy in {(oy - 1) >> 1 (floor), that + 1}
    const int iy0 = (oy + 1) / 2 - 1, ix0 = (ox + 1) / 2 - 1;
    float a = 0.f;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const int iy = iy0 + dy, ix = ix0 + dx;
        if (iy < 0 || iy >= Hi || ix < 0 || ix >= Wi) continue;
        const int ky = oy + 1 - 2 * iy, kx = ox + 1 - 2 * ix;
        const float wy = (ky == 0 || ky == 3) ? 0.25f : 0.75f, wx = (kx == 0 || kx == 3) ? 0.25f : 0.75f;
        a = fmaf(x[(((int64_t)n * Hi + iy) * Wi + ix) * C + c], wy * wx, a);
      }
    if (scale != nullptr) a = fmaf(a, scale[c], shift[c]);
    y[i] = fmaxf(a, 0.f) + (res != nullptr ? res[i] : 0.f);
  }
}

// The decoder head WITHOUT the commutation (a batch norm with a shift sits between the x8 deconv and its relu, so the class
// scores cannot be interpolated at 1/8 resolution): per output pixel the 16x16 / stride-8 bilinear transposed conv of the U
// feature channels (two source rows x two source columns; taps w1[k] = 1 - |k - 7.5| / 8, custom_layers.py:8-25), the affine,
// the relu, the 1x1 score conv, bias, softmax, argmax -- all in fp32 (simple_fcn.py:89-135 on float32).  One thread per pixel.
__global__ __launch_bounds__(256) void decoder_head_affine_f32_kernel(const float* __restrict__ f, const float* __restrict__ scale,
                                                                     const float* __restrict__ shift, const float* __restrict__ ws,
                                                                     const float* __restrict__ bs, int N, int Hi, int Wi, int U, int C,
                                                                     float* __restrict__ score, float* __restrict__ prob,
                                                                     int64_t* __restrict__ label) {
  const int Ho = 8 * Hi, Wo = 8 * Wi;
  const int64_t total = (int64_t)N * Ho * Wo;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int ox = (int)(i % Wo);
    int64_t r = i / Wo;
    const int oy = (int)(r % Ho);
    const int n = (int)(r / Ho);
    // sources iy with 0 <= oy + 4 - 8 iy < 16
    const int iy0 = (oy + 4) / 8 - 1, ix0 = (ox + 4) / 8 - 1;
    float wgt[4];
    const float* src[4];
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const int iy = iy0 + dy, ix = ix0 + dx;
        const bool in = iy >= 0 && iy < Hi && ix >= 0 && ix < Wi;
        const int ky = oy + 4 - 8 * iy, kx = ox + 4 - 8 * ix;
        wgt[2 * dy + dx] = in ? (1.f - fabsf((float)ky - 7.5f) * 0.125f) * (1.f - fabsf((float)kx - 7.5f) * 0.125f) : 0.f;
        src[2 * dy + dx] = f + (((int64_t)n * Hi + (in ? iy : 0)) * Wi + (in ? ix : 0)) * U;
      }
    float s[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) s[k] = k < C ? bs[k] : 0.f;
    for (int u = 0; u < U; ++u) {
      float a = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t) a = fmaf(src[t][u], wgt[t], a);
      a = fmaxf(fmaf(a, scale[u], shift[u]), 0.f);
#pragma unroll
      for (int k = 0; k < 32; ++k)
        if (k < C) s[k] = fmaf(a, ws[u * C + k], s[k]);
    }
    float m = s[0];
    int best = 0;
#pragma unroll
    for (int k = 1; k < 32; ++k)
      if (k < C && s[k] > m) {
        m = s[k];
        best = k;
      }
    if (score != nullptr)
      for (int k = 0; k < C; ++k) score[i * C + k] = s[k];
    if (prob != nullptr) {
      float e[32], sum = 0.f;
#pragma unroll
      for (int k = 0; k < 32; ++k) {
        e[k] = k < C ? expf(s[k] - m) : 0.f;
        sum += e[k];
      }
      for (int k = 0; k < C; ++k) prob[i * C + k] = e[k] / sum;
    }
    if (label != nullptr) label[i] = best;
  }
}

__global__ __launch_bounds__(256) void score_lowres_f32_kernel(const float* __restrict__ f, const float* __restrict__ ws,
                                                              float* __restrict__ S, int N, int Hi, int Wi, int U, int C,
                                                              int CP) {
  const int64_t total = (int64_t)N * Hi * Wi * CP;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int k = (int)(i % CP);
    int64_t r = i / CP;
    const int x = (int)(r % Wi);
    r /= Wi;
    const int y = (int)(r % Hi);
    const int n = (int)(r / Hi);
    float a = 0.f;
    if (k < C) {
      const float* src = f + (((int64_t)n * Hi + y) * Wi + x) * U;
      for (int u = 0; u < U; ++u) a = fmaf(src[u], ws[u * C + k], a);
    }
    S[(((int64_t)n * (Hi + 2) + y + 1) * (Wi + 2) + x + 1) * CP + k] = a;
  }
}

inline unsigned f32_grid(int64_t total) {
  int64_t g = (total + 255) / 256;
  return (unsigned)(g < 1 ? 1 : (g > 65535 ? 65535 : g));
}

}  // namespace

extern "C" int xv_conv2d_f32_scalar(const float* x, int n, int h, int w, int cin, const float* w_hwio, const float* bias,
                                    int k, int cout, int relu, float* y, void* stream) {
  XV_CHECK_ARG(x && w_hwio && bias && y);
  XV_CHECK_SHAPE(n > 0 && h > 0 && w > 0 && cin > 0 && cout > 0 && (k == 1 || k == 3));
  const int64_t tiles = (int64_t)n * ((h + 7) / 8) * ((w + 7) / 8);
  XV_CHECK_SHAPE(tiles <= 0x7fffffff && (cout + 63) / 64 <= 65535);
  const dim3 grid((unsigned)tiles, (unsigned)((cout + 63) / 64));
  if (k == 3)
    hipLaunchKernelGGL(conv_f32_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, x, w_hwio, bias, y, n, h, w, cin, cout, relu);
  else
    hipLaunchKernelGGL(conv_f32_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, x, w_hwio, bias, y, n, h, w, cin, cout, relu);
  return xv_launch_status();
}

template <int KS, int CK, bool VEC>
static int launch_conv_f32_mfma(const float* x, int n, int h, int w, int cin, const float* w_hwio, const float* bias, int cout,
                                int relu, float* y, float* pooled, hipStream_t stream) {
  using G = F32Geo<KS, CK>;
  static bool lds_ok[XV_MAX_DEVICES] = {false};
  hipError_t e = xv_allow_dynamic_lds((const void*)conv_f32_mfma_kernel<KS, CK, VEC>, G::LDS_BYTES, lds_ok, false);
  if (e != hipSuccess) return (int)e;
  const int tiles_x = (w + G::TW - 1) / G::TW, tiles_y = (h + G::TH - 1) / G::TH, ncb = (cout + 63) / 64;
  const int64_t total = (int64_t)n * tiles_x * tiles_y * ncb;
  XV_CHECK_SHAPE(total <= 0x7fffffff);
  // persistent grid: as many workgroups as the chip holds at once (LDS-limited; at most 3 per CU), each walking tiles b, b + grid, ..
  int per_cu = 160 * 1024 / G::LDS_BYTES;
  per_cu = per_cu < 1 ? 1 : (per_cu > 3 ? 3 : per_cu);
  const int64_t resident = (int64_t)xv_num_cus() * per_cu;
  const unsigned grid = (unsigned)(total < resident ? total : resident);
  hipLaunchKernelGGL((conv_f32_mfma_kernel<KS, CK, VEC>), dim3(grid), dim3(256), G::LDS_BYTES, stream, x, w_hwio, bias, y, pooled,
                     h, w, cin, cout, relu, tiles_x, tiles_y, ncb, (int)total);
  return xv_launch_status();
}

extern "C" int xv_conv2d_f32_pool(const float* x, int n, int h, int w, int cin, const float* w_hwio, const float* bias, int k,
                                  int cout, int relu, float* y, float* pooled, void* stream) {
  XV_CHECK_ARG(x && w_hwio && bias && (y || pooled));
  XV_CHECK_SHAPE(n > 0 && h > 0 && w > 0 && cin > 0 && cout > 0 && (k == 1 || k == 3));
  XV_CHECK_SHAPE(pooled == nullptr || ((h & 1) == 0 && (w & 1) == 0));
  // image-relative and weight offsets are 32-bit byte counts below 2^31 in the kernel (buffer descriptors)
  XV_CHECK_SHAPE((int64_t)h * w * (cin > cout ? cin : cout) < ((int64_t)1 << 29) && (int64_t)k * k * cin * cout < ((int64_t)1 << 29));
  hipStream_t st = (hipStream_t)stream;
  if (cin <= 4)        // conv1_1: chunks of four channels
    return k == 3 ? launch_conv_f32_mfma<3, 4, false>(x, n, h, w, cin, w_hwio, bias, cout, relu, y, pooled, st)
                  : launch_conv_f32_mfma<1, 4, false>(x, n, h, w, cin, w_hwio, bias, cout, relu, y, pooled, st);
  // 16-byte patch loads need whole channel quads at 16-byte aligned addresses
  const bool vec = (cin & 3) == 0 && ((uintptr_t)x & 15) == 0;
  if (k == 3)
    return vec ? launch_conv_f32_mfma<3, 16, true>(x, n, h, w, cin, w_hwio, bias, cout, relu, y, pooled, st)
               : launch_conv_f32_mfma<3, 16, false>(x, n, h, w, cin, w_hwio, bias, cout, relu, y, pooled, st);
  return vec ? launch_conv_f32_mfma<1, 16, true>(x, n, h, w, cin, w_hwio, bias, cout, relu, y, pooled, st)
             : launch_conv_f32_mfma<1, 16, false>(x, n, h, w, cin, w_hwio, bias, cout, relu, y, pooled, st);
}

extern "C" int xv_conv2d_f32(const float* x, int n, int h, int w, int cin, const float* w_hwio, const float* bias, int k,
                             int cout, int relu, float* y, void* stream) {
  XV_CHECK_ARG(y);
  return xv_conv2d_f32_pool(x, n, h, w, cin, w_hwio, bias, k, cout, relu, y, nullptr, stream);
}

extern "C" int xv_maxpool2x2_f32(const float* x, int n, int h, int w, int c, float* y, void* stream) {
  XV_CHECK_ARG(x && y);
  XV_CHECK_SHAPE(n > 0 && h > 0 && w > 0 && c > 0 && (h & 1) == 0 && (w & 1) == 0);
  const int64_t total = (int64_t)n * (h / 2) * (w / 2) * c;
  hipLaunchKernelGGL(maxpool_f32_kernel, dim3(f32_grid(total)), dim3(256), 0, (hipStream_t)stream, x, y, n, h / 2, w / 2, c);
  return xv_launch_status();
}

extern "C" int xv_upsample2x_f32(const float* x, int n, int h, int w, int c, const float* residual, float* y, void* stream) {
  XV_CHECK_ARG(x && y);
  XV_CHECK_SHAPE(n > 0 && h > 0 && w > 0 && c > 0);
  const int64_t total = (int64_t)n * 4 * h * w * c;
  hipLaunchKernelGGL(upsample2x_f32_kernel, dim3(f32_grid(total)), dim3(256), 0, (hipStream_t)stream, x, residual, y, n, h, w, c,
                     (const float*)nullptr, (const float*)nullptr);
  return xv_launch_status();
}

// y = relu(bilinear_x2(x) * scale + shift) + residual (the batch norm of upscore_conv5 in float32)
extern "C" int xv_upsample2x_affine_f32(const float* x, int n, int h, int w, int c, const float* scale, const float* shift,
                                        const float* residual, float* y, void* stream) {
  XV_CHECK_ARG(x && y && scale && shift);
  XV_CHECK_SHAPE(n > 0 && h > 0 && w > 0 && c > 0);
  const int64_t total = (int64_t)n * 4 * h * w * c;
  hipLaunchKernelGGL(upsample2x_f32_kernel, dim3(f32_grid(total)), dim3(256), 0, (hipStream_t)stream, x, residual, y, n, h, w, c,
                     scale, shift);
  return xv_launch_status();
}

// The un-commuted decoder head in float32 (decoder_head_affine_f32_kernel): fused [n][h][w][u] -> score / prob
// [n][8h][8w][C] float32 and label [n][8h][8w] int64 (any of them may be null)
extern "C" int xv_decoder_head_affine_f32(const float* fused, int n, int h, int w, int u, const float* scale, const float* shift,
                                          const float* w_score, const float* b_score, int num_classes, float* score, float* prob,
                                          int64_t* label, void* stream) {
  XV_CHECK_ARG(fused && scale && shift && w_score && b_score && (score || prob || label));
  XV_CHECK_SHAPE(n > 0 && h > 0 && w > 0 && u > 0 && num_classes >= 1 && num_classes <= 32);
  const int64_t total = (int64_t)n * 64 * h * w;
  hipLaunchKernelGGL(decoder_head_affine_f32_kernel, dim3(f32_grid(total)), dim3(256), 0, (hipStream_t)stream, fused, scale, shift,
                     w_score, b_score, n, h, w, u, num_classes, score, prob, label);
  return xv_launch_status();
}

extern "C" int xv_score_lowres_f32(const float* fused, int n, int h, int w, int u, const float* w_score, int num_classes,
                                   float* S, void* stream) {
  XV_CHECK_ARG(fused && w_score && S);
  XV_CHECK_SHAPE(n > 0 && h > 0 && w > 0 && u > 0 && num_classes >= 1 && num_classes <= 32);
  const int cp = (num_classes + 3) / 4 * 4;
  const int64_t total = (int64_t)n * h * w * cp;
  hipLaunchKernelGGL(score_lowres_f32_kernel, dim3(f32_grid(total)), dim3(256), 0, (hipStream_t)stream, fused, w_score, S, n,
                     h, w, u, num_classes, cp);
  return xv_launch_status();
}
