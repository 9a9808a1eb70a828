// Backward pass of the HBM-bound FCN layers, the softmax cross-entropy head, and the optimizers
// (gfx950).  Together with conv_wgrad.hip and the dgrad mode of conv_mfma.hip these replace the
// gradient ops that tf.train.{Adam,RMSProp,Adagrad}Optimizer.minimize(self.loss) builds over the
// training graph of SimpleFCN (base_model.py:153-162, simple_fcn.py:200-214).
#include <stdlib.h>

#include "xv_common.h"

namespace {

inline int grid_for(int64_t total, int per_block = 256, int cap = 8192) {
  int64_t g = (total + per_block - 1) / per_block;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

__device__ inline float bf_lo(uint32_t w) { return bf16_bits_to_f32(w & 0xffffu); }
__device__ inline float bf_hi(uint32_t w) { return __builtin_bit_cast(float, w & 0xffff0000u); }

// ---- MaxPoolGrad + ReluGrad: dy[pos] = dpooled if pos is the (first) max of its 2x2 window and
// y[pos] > 0, else 0  (max_pooling2d, simple_fcn.py:41,44,48,58 under the relu of the conv above) ----
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const __bf16* __restrict__ y, const __bf16* __restrict__ dp,
                                                         __bf16* __restrict__ dy, int N, int Ho, int Wo, int C) {
  const int c8 = C >> 3;
  const int64_t total = (int64_t)N * Ho * Wo * c8;
  const int Hi = Ho * 2, Wi = Wo * 2;
  const int64_t rowp = (int64_t)(Wi + 2) * C;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int cg = (int)(idx % c8);
    int64_t r = idx / c8;
    const int ox = (int)(r % Wo);
    r /= Wo;
    const int oy = (int)(r % Ho);
    const int n = (int)(r / Ho);
    const int64_t off = (((int64_t)n * (Hi + 2) + (2 * oy + 1)) * (Wi + 2) + (2 * ox + 1)) * C + cg * 8;
    const u32x4 g = *reinterpret_cast<const u32x4*>(dp + (((int64_t)n * (Ho + 2) + (oy + 1)) * (Wo + 2) + (ox + 1)) * C + cg * 8);
    u32x4 v[4] = {*reinterpret_cast<const u32x4*>(y + off), *reinterpret_cast<const u32x4*>(y + off + C),
                  *reinterpret_cast<const u32x4*>(y + off + rowp), *reinterpret_cast<const u32x4*>(y + off + rowp + C)};
    u32x4 o[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float val[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) val[k] = h ? bf_hi(v[k][w]) : bf_lo(v[k][w]);
        int best = 0;
        float m = val[0];
#pragma unroll
        for (int k = 1; k < 4; ++k)
          if (val[k] > m) {
            m = val[k];
            best = k;
          }
        const uint32_t gb = h ? (g[w] & 0xffff0000u) : (g[w] & 0xffffu);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const uint32_t bits = (k == best && m > 0.f) ? gb : 0u;
          o[k][w] = h ? (o[k][w] | bits) : bits;
        }
      }
    }
    *reinterpret_cast<u32x4*>(dy + off) = o[0];
    *reinterpret_cast<u32x4*>(dy + off + C) = o[1];
    *reinterpret_cast<u32x4*>(dy + off + rowp) = o[2];
    *reinterpret_cast<u32x4*>(dy + off + rowp + C) = o[3];
  }
}

// ---- ReluGrad on padded-NHWC bf16: out = ref > 0 ? g : 0 ---------------------------------------
__global__ __launch_bounds__(256) void relu_bwd_kernel(const __bf16* __restrict__ g, const __bf16* __restrict__ ref,
                                                      __bf16* __restrict__ out, int64_t n8) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    const u32x4 a = *reinterpret_cast<const u32x4*>(g + i * 8);
    const u32x4 r = *reinterpret_cast<const u32x4*>(ref + i * 8);
    u32x4 o;
#pragma unroll
    for (int w = 0; w < 4; ++w)
      o[w] = (bf_lo(r[w]) > 0.f ? (a[w] & 0xffffu) : 0u) | (bf_hi(r[w]) > 0.f ? (a[w] & 0xffff0000u) : 0u);
    *reinterpret_cast<u32x4*>(out + i * 8) = o;
  }
}

template <int S>
__device__ inline float bilinear_w(int o, int i) {
  // weight with which source index i feeds output o (conv2d_transpose 'same', k = 2S): 0 if not a neighbour
  const int t = o + S / 2;
  const int i1 = t / S, p1 = t - i1 * S;
  constexpr float center = (2.f * S - 1.f - (S % 2)) / (2.f * S);
  if (i == i1) return 1.f - fabsf((float)p1 / S - center);
  if (i == i1 - 1) return 1.f - fabsf((float)(p1 + S) / S - center);
  return 0.f;
}

// ---- backward of fused = s4 + relu(bilinear_x2(s5))  w.r.t. s5, through the relu of score_conv5 ----
// ds5[i,j,c] = (s5 > 0) * sum_{(oy,ox) in the 4x4 footprint} wy*wx * dfused[oy,ox,c] * (up2(s5)[oy,ox,c] > 0)
__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(const __bf16* __restrict__ df, const __bf16* __restrict__ s5,
                                                            __bf16* __restrict__ ds5, int N, int Hi, int Wi, int C) {
  const int c8 = C >> 3;
  const int Ho = 2 * Hi, Wo = 2 * Wi;
  const int64_t total = (int64_t)N * Hi * Wi * c8;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int cg = (int)(idx % c8);
    int64_t r = idx / c8;
    const int j = (int)(r % Wi);
    r /= Wi;
    const int i = (int)(r % Hi);
    const int n = (int)(r / Hi);
    const __bf16* simg = s5 + (int64_t)n * (Hi + 2) * (Wi + 2) * C + cg * 8;
    const __bf16* dimg = df + (int64_t)n * (Ho + 2) * (Wo + 2) * C + cg * 8;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int oy = 2 * i - 1; oy <= 2 * i + 2; ++oy) {
      if (oy < 0 || oy >= Ho) continue;
      const float wy = bilinear_w<2>(oy, i);
      const int ty = oy + 1, iy1 = ty >> 1;
      const float wy1 = bilinear_w<2>(oy, iy1), wy0 = bilinear_w<2>(oy, iy1 - 1);
      for (int ox = 2 * j - 1; ox <= 2 * j + 2; ++ox) {
        if (ox < 0 || ox >= Wo) continue;
        const float wx = bilinear_w<2>(ox, j);
        const int tx = ox + 1, ix1 = tx >> 1;
        const float wx1 = bilinear_w<2>(ox, ix1), wx0 = bilinear_w<2>(ox, ix1 - 1);
        // recompute the forward value up2(s5)[oy,ox] for the relu mask (padded coords: logical i -> i+1)
        const __bf16* p00 = simg + ((int64_t)iy1 * (Wi + 2) + ix1) * C;
        const int64_t rowp = (int64_t)(Wi + 2) * C;
        const u32x4 a00 = *reinterpret_cast<const u32x4*>(p00), a01 = *reinterpret_cast<const u32x4*>(p00 + C);
        const u32x4 a10 = *reinterpret_cast<const u32x4*>(p00 + rowp), a11 = *reinterpret_cast<const u32x4*>(p00 + rowp + C);
        const u32x4 gv = *reinterpret_cast<const u32x4*>(dimg + ((int64_t)(oy + 1) * (Wo + 2) + (ox + 1)) * C);
        const float w00 = wy0 * wx0, w01 = wy0 * wx1, w10 = wy1 * wx0, w11 = wy1 * wx1;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const float u_lo = bf_lo(a00[w]) * w00 + bf_lo(a01[w]) * w01 + bf_lo(a10[w]) * w10 + bf_lo(a11[w]) * w11;
          const float u_hi = bf_hi(a00[w]) * w00 + bf_hi(a01[w]) * w01 + bf_hi(a10[w]) * w10 + bf_hi(a11[w]) * w11;
          if (u_lo > 0.f) acc[2 * w] += wy * wx * bf_lo(gv[w]);
          if (u_hi > 0.f) acc[2 * w + 1] += wy * wx * bf_hi(gv[w]);
        }
      }
    }
    const int64_t so = ((int64_t)(i + 1) * (Wi + 2) + (j + 1)) * C;
    const u32x4 sv = *reinterpret_cast<const u32x4*>(simg + so);
    u32x4 o;
#pragma unroll
    for (int w = 0; w < 4; ++w)
      o[w] = pack_bf16x2(bf_lo(sv[w]) > 0.f ? acc[2 * w] : 0.f, bf_hi(sv[w]) > 0.f ? acc[2 * w + 1] : 0.f);
    *reinterpret_cast<u32x4*>(ds5 + (int64_t)n * (Hi + 2) * (Wi + 2) * C + cg * 8 + so) = o;
  }
}

// ---- number of labelled pixels: sum(one_hot(labels)) of utils.py:52 --------------------------------
__global__ __launch_bounds__(1024) void count_valid_kernel(const int32_t* __restrict__ labels, int C, int64_t npix,
                                                         unsigned long long* __restrict__ count) {
  unsigned int c = 0;
  // four labels per 16-byte load, the last npix % 4 alone
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  const bool vec = ((uintptr_t)labels & 15) == 0;
  const int64_t nquads = vec ? npix >> 2 : 0;
  for (int64_t q = (int64_t)blockIdx.x * 1024 + threadIdx.x; q < nquads; q += (int64_t)gridDim.x * 1024) {
    const i32x4 l = *reinterpret_cast<const i32x4*>(labels + 4 * q);
#pragma unroll
    for (int j = 0; j < 4; ++j) c += ((unsigned)l[j] < (unsigned)C) ? 1u : 0u;
  }
  for (int64_t i = 4 * nquads + (int64_t)blockIdx.x * 1024 + threadIdx.x; i < npix; i += (int64_t)gridDim.x * 1024) {
    const int l = labels[i];
    c += (l >= 0 && l < C) ? 1u : 0u;
  }
  __shared__ unsigned int s;
  if (threadIdx.x == 0) s = 0;
  __syncthreads();
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) c += __shfl_xor(c, off, 64);
  if ((threadIdx.x & 63) == 0) atomicAdd(&s, c);
  __syncthreads();
  if (threadIdx.x == 0 && s) atomicAdd(count, (unsigned long long)s);
}

// ---- head backward (loss + gradients), in the same commuted form as the forward head ---------------
// score = bilinear_x8(S) + bs with S = fused . Ws at 1/8 resolution (see pointwise.hip): the relu of the
// x8 deconv is the identity because fused >= 0, and wherever it is not strictly the identity (all four
// source features of a channel are 0) the gradient entries that differ are zeroed again by the relu
// masks of score_conv4 / upscore_conv5 further down, so the linear form gives the reference's gradients.
// Kernel 1, per COLUMN of eight output pixels (rows 8 ib .. 8 ib + 7 of column ox): interpolate S, softmax,
//   loss += -log p[label]/denom, dscore = (p*valid - onehot)/denom; bias gradient by block reduction.  dscore itself
//   never reaches memory: the x8 bilinear deconv's gradient is separable, and the column's eight pixels feed only the
//   1/8-resolution rows ib-1, ib, ib+1 -- the thread keeps those three row-weighted sums (P[n][ib][slot][ox][CM], slot =
//   target row - (ib - 1)) and kernel 2 finishes along x.  (Round 1 wrote the dense fp32 gradient, 226 MB at 16 images,
//   and kernel 2 gathered each 1/8-resolution pixel's 16 x 16 footprint from it: 390 us for the pair.)
// FULL: the class count is CM (every `k < C` folds away).  exp / log / reciprocal are the hardware's (xv_common.h, as in the
// inference heads: relative error ~1e-7, far inside what a bf16 training step resolves).
template <int CM, bool FULL = false>
__global__ __launch_bounds__(256) void head_loss_kernel(const float* __restrict__ S, const float* __restrict__ bs_g,
                                                       const int32_t* __restrict__ labels,
                                                       const unsigned long long* __restrict__ count, int N, int Hi,
                                                       int Wi, int C_, double* __restrict__ loss, float* __restrict__ dbs,
                                                       float* __restrict__ P, float* __restrict__ part_out) {
  const int C = FULL ? CM : C_;
  __shared__ float red[4][CM + 1];
  const int Ho = Hi * 8, Wo = Wi * 8;
  const int64_t ncols = (int64_t)N * Hi * Wo;
  float part[CM], lossterm = 0.f;
#pragma unroll
  for (int k = 0; k < CM; ++k) part[k] = 0.f;
  const float inv_denom = 1.f / (1e-20f + (float)(*count));
  // grid-stride over the columns with the sums kept in registers: the 13 global atomics at the end are issued once
  // per workgroup of a bounded grid (same-address atomics serialise at the memory side)
  for (int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x; col < ncols; col += (int64_t)gridDim.x * 256) {
    const int c32 = (int)col;  // ncols < 2^31 (checked by the launcher): 32-bit divisions
    const int crow = c32 / Wo;  // n * Hi + ib
    const int ox = c32 - crow * Wo, n = crow / Hi, ib = crow - n * Hi;
    const int ix1 = (ox + 4) >> 3;
    const float wx1 = bilinear_w<8>(ox, ix1), wx0 = bilinear_w<8>(ox, ix1 - 1);
    const int64_t rowp = (int64_t)(Wi + 2) * CM;
    float colsum[3][CM];
#pragma unroll
    for (int sl = 0; sl < 3; ++sl)
#pragma unroll
      for (int k = 0; k < CM; ++k) colsum[sl][k] = 0.f;
    // The column's eight pixels interpolate between the padded source rows ib, ib + 1 (r < 4) and ib + 1, ib + 2 (r >= 4) at
    // columns ix1, ix1 + 1: 6 CM / 4 loads, made once.  (Written per pixel -- four loads per class quad in each of the
    // eight rows -- the compiler requested all 96 first: 256 VGPRs + 156 AGPRs, one wave per SIMD.)
    f32x4 T[3][2][CM / 4];
    {
      const float* t00 = S + (((int64_t)n * (Hi + 2) + ib) * (Wi + 2) + ix1) * CM;
#pragma unroll
      for (int sr = 0; sr < 3; ++sr)
#pragma unroll
        for (int sx = 0; sx < 2; ++sx)
#pragma unroll
          for (int q = 0; q < CM / 4; ++q) T[sr][sx][q] = *reinterpret_cast<const f32x4*>(t00 + sr * rowp + sx * CM + 4 * q);
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int oy = 8 * ib + r;
      const int iy1 = ib + (r >= 4 ? 1 : 0);  // = (oy + 4) >> 3
      const float wy1 = bilinear_w<8>(oy, iy1), wy0 = bilinear_w<8>(oy, iy1 - 1);
      const float w00 = wy0 * wx0, w01 = wy0 * wx1, w10 = wy1 * wx0, w11 = wy1 * wx1;
      float sc[CM];
#pragma unroll
      for (int k4 = 0; k4 < CM; k4 += 4) {
        const f32x4 a = T[r >= 4 ? 1 : 0][0][k4 / 4], b = T[r >= 4 ? 1 : 0][1][k4 / 4];
        const f32x4 c = T[r >= 4 ? 2 : 1][0][k4 / 4], d = T[r >= 4 ? 2 : 1][1][k4 / 4];
        const f32x4 v = a * w00 + b * w01 + c * w10 + d * w11;
        sc[k4] = v.x;
        sc[k4 + 1] = v.y;
        sc[k4 + 2] = v.z;
        sc[k4 + 3] = v.w;
      }
#pragma unroll
      for (int k = 0; k < CM; ++k) sc[k] += bs_g[k < C ? k : C - 1];
      const int lab = labels[((int64_t)n * Ho + oy) * Wo + ox];
      const bool valid = lab >= 0 && lab < C;
      float m = sc[0];
#pragma unroll
      for (int k = 1; k < CM; ++k)
        if (k < C) m = fmaxf(m, sc[k]);
      float sum = 0.f, zlab = 0.f;
#pragma unroll
      for (int k = 0; k < CM; ++k) {
        if (k == lab) zlab = sc[k] - m;
        sc[k] = k < C ? xv_fast_exp(sc[k] - m) : 0.f;
        sum += sc[k];
      }
      const float rsum = xv_fast_rcp(sum);  // one reciprocal per pixel (the gradient does not need the forward's exact quotients)
#pragma unroll
      for (int k = 0; k < CM; ++k) {
        const float p = sc[k] * rsum;
        sc[k] = (valid && k < C) ? (p - (k == lab ? 1.f : 0.f)) * inv_denom : 0.f;
      }
      lossterm += valid ? -(zlab - xv_fast_log(sum)) * inv_denom : 0.f;  // -(log_softmax)[label] / denom
      // this row feeds 1/8-resolution rows iy1 - 1 (weight wy0) and iy1 (weight wy1): slots r < 4 ? (0, 1) : (1, 2)
#pragma unroll
      for (int k = 0; k < CM; ++k) {
        part[k] += sc[k];  // already zero for unlabelled pixels
        colsum[r >= 4 ? 1 : 0][k] = fmaf(wy0, sc[k], colsum[r >= 4 ? 1 : 0][k]);
        colsum[r >= 4 ? 2 : 1][k] = fmaf(wy1, sc[k], colsum[r >= 4 ? 2 : 1][k]);
      }
    }
#pragma unroll
    for (int sl = 0; sl < 3; ++sl) {
      float* dst = P + (((int64_t)crow * 3 + sl) * Wo + ox) * CM;
#pragma unroll
      for (int k4 = 0; k4 < CM; k4 += 4)
        *reinterpret_cast<f32x4*>(dst + k4) = f32x4{colsum[sl][k4], colsum[sl][k4 + 1], colsum[sl][k4 + 2], colsum[sl][k4 + 3]};
    }
  }
  // bias gradient and loss: butterfly sum over the wave, then ONE LDS atomic per wave and channel (per-lane LDS
  // atomics on 13 shared addresses serialise 64-fold and used to be most of this kernel's time)
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    lossterm += __shfl_xor(lossterm, off, 64);
#pragma unroll
    for (int k = 0; k < CM; ++k) part[k] += __shfl_xor(part[k], off, 64);
  }
  // (the butterfly leaves the same bits in every lane: xor exchanges add the two halves in both orders, a + b == b + a)
  if ((threadIdx.x & 63) == 0) {
    red[threadIdx.x >> 6][CM] = lossterm;
#pragma unroll
    for (int k = 0; k < CM; ++k) red[threadIdx.x >> 6][k] = part[k];
  }
  __syncthreads();
  if (threadIdx.x <= CM) {
    // the four waves in a fixed order; then this workgroup's partial sums go to its own row of `part_out`, which
    // partials_reduce_kernel adds up in workgroup order (bitwise reproducible loss and bias gradient) -- or, without
    // that buffer, straight into the results with atomics
    const float v = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
    if (part_out != nullptr)
      part_out[(int64_t)blockIdx.x * (CM + 1) + threadIdx.x] = v;
    else if (threadIdx.x < C && v != 0.f)
      atomicAdd(&dbs[threadIdx.x], v);
    else if (threadIdx.x == CM && v != 0.f)
      atomicAdd(loss, (double)v);
  }
}

// out[i] += sum_b part[b][i] for i < n_out, out64 += sum_b part[b][i64] (a loss): LPE lanes per element each take every
// LPE-th row, then a butterfly (and, for LPE = 256, the four waves' sums in wave order through LDS) -- a fixed tree, so the
// sums are bitwise reproducible (as head_dws_reduce_kernel).  LPE = 256 (one workgroup per element) for the short vectors
// (the head's 13 sums over 4 096 rows, the 64 bias sums): with 16 lanes per element those ran 128-256 dependent adds per
// lane in a single workgroup, 23-37 us per launch.
template <int LPE>
__global__ __launch_bounds__(256) void partials_reduce_kernel(const float* __restrict__ part, int nblocks, int len,
                                                             float* __restrict__ out, int n_out, double* __restrict__ out64,
                                                             int i64) {
  static_assert(LPE == 64 || LPE == 256, "a wave or a workgroup per element");
  __shared__ double wsum[4];
  const int i = blockIdx.x * (256 / LPE) + threadIdx.x / LPE, sub = threadIdx.x % LPE;
  const bool live = i < len;
  double a = 0.0;  // (fp32 terms in a double accumulator: exact enough for the loss, rounded once for the gradients)
  if (live)
    for (int b = sub; b < nblocks; b += LPE) a += (double)part[(int64_t)b * len + i];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) a += __shfl_xor(a, off, 64);
  if constexpr (LPE == 256) {
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = a;
    __syncthreads();
    a = ((wsum[0] + wsum[1]) + wsum[2]) + wsum[3];
  }
  if (!live || sub != 0) return;
  if (i < n_out)
    out[i] += (float)a;
  else if (out64 != nullptr && i == i64)
    *out64 += a;
}

// Kernel 2, per 1/8-resolution pixel: dS[k] = sum over the 16 footprint columns of wx * (kernel 1's row sums)
// (gradient of the x8 bilinear deconv), dfused[u] = sum_k dS[k]*Ws[u][k], and the score-weight
// gradient dWs[u][k] += fused[u]*dS[k] reduced over the workgroup's 256 pixels through LDS.
// FOUR threads per pixel (1,024-thread workgroups; a 1/8-resolution map has only 74 k pixels at 16 images): thread q of
// a pixel sums footprint columns 4q .. 4q+3, the four partial sums meet through two lane exchanges; then it produces
// units 8q .. 8q+7 (+32 ...) of dfused, and the workgroup's 768 weight-gradient cells get one thread each.
template <int CM>
__global__ __launch_bounds__(1024) void head_bwd_lowres_kernel(const float* __restrict__ P,
                                                              const __bf16* __restrict__ f,
                                                              const float* __restrict__ ws_g, int N, int Hi, int Wi,
                                                              int U, int C, float* __restrict__ dws_part,
                                                              __bf16* __restrict__ df) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* wsm = sm;                                         // [U][CM]
  float* dsm = wsm + U * CM;                               // [256][CM]
  __bf16* fm = reinterpret_cast<__bf16*>(dsm + 256 * CM);  // [256][U]
  for (int i = threadIdx.x; i < U * CM; i += 1024) {
    const int u = i / CM, k = i - u * CM;
    wsm[i] = k < C ? ws_g[u * C + k] : 0.f;
  }
  const int Wo = 8 * Wi;
  const int64_t total = (int64_t)N * Hi * Wi;
  const int lp = threadIdx.x >> 2, q = threadIdx.x & 3;  // pixel of the workgroup, quarter
  const int64_t idx = (int64_t)blockIdx.x * 256 + lp;
  const bool live = idx < total;
  float ds[CM];
#pragma unroll
  for (int k = 0; k < CM; ++k) ds[k] = 0.f;
  int64_t pad_off = 0;
  if (live) {
    const int j = (int)(idx % Wi);
    const int i = (int)((idx / Wi) % Hi);
    const int n = (int)(idx / ((int64_t)Wi * Hi));
    // along x over kernel 1's row sums: 1/8-resolution row i collects slot 1 of its own column block, slot 0 of the
    // block below and slot 2 of the block above; this thread takes 4 of the 16 footprint columns
    // branch-free: a column / row block outside the map is read at a clamped (valid, finite) position with weight 0 --
    // fmaf(0, g, ds) = ds exactly (ds is never -0) -- so that a column's 3 CM / 4 loads are requested together and the next
    // column's may follow without waiting for a range check
    // (one column = 3 CM / 4 vectors at a time: more in flight would not fit the 128 registers of a 1 024-thread workgroup)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      f32x4 g[3][CM / 4];
      float wv[3];
      const int ox = 8 * j - 4 + 4 * q + c;
      const bool okx = ox >= 0 && ox < Wo;
      const int oxc = ox < 0 ? 0 : (ox >= Wo ? Wo - 1 : ox);
      const float w = bilinear_w<8>(oxc, j);
#pragma unroll
      for (int sl = 0; sl < 3; ++sl) {
        const int ib = i + 1 - sl;  // the column block whose slot sl targets row i
        const bool ok = okx && ib >= 0 && ib < Hi;
        const int ibc = ib < 0 ? 0 : (ib >= Hi ? Hi - 1 : ib);
        wv[sl] = ok ? w : 0.f;
        const float* src = P + ((((int64_t)n * Hi + ibc) * 3 + sl) * Wo + oxc) * CM;
#pragma unroll
        for (int k4 = 0; k4 < CM / 4; ++k4) g[sl][k4] = *reinterpret_cast<const f32x4*>(src + 4 * k4);
      }
#pragma unroll
      for (int sl = 0; sl < 3; ++sl)
#pragma unroll
        for (int k4 = 0; k4 < CM / 4; ++k4) {
          const float ww = wv[sl];
          ds[4 * k4] = fmaf(ww, g[sl][k4].x, ds[4 * k4]);
          ds[4 * k4 + 1] = fmaf(ww, g[sl][k4].y, ds[4 * k4 + 1]);
          ds[4 * k4 + 2] = fmaf(ww, g[sl][k4].z, ds[4 * k4 + 2]);
          ds[4 * k4 + 3] = fmaf(ww, g[sl][k4].w, ds[4 * k4 + 3]);
        }
      if (c & 1) asm volatile("" ::: "memory");  // at most two columns' loads in flight
    }
    pad_off = (((int64_t)n * (Hi + 2) + (i + 1)) * (Wi + 2) + (j + 1)) * U;
  }
  // the four quarters of a pixel sit in adjacent lanes: after the two exchanges every one of them holds the sum
#pragma unroll
  for (int k = 0; k < CM; ++k) {
    ds[k] += __shfl_xor(ds[k], 1);
    ds[k] += __shfl_xor(ds[k], 2);
  }
  if (q == 0) {
#pragma unroll
    for (int k = 0; k < CM; ++k) dsm[lp * CM + k] = ds[k];
  }
  __syncthreads();  // wsm ready
  for (int u0 = q * 8; u0 < U; u0 += 32) {
    u32x4 fv = u32x4{0u, 0u, 0u, 0u};
    if (live) fv = *reinterpret_cast<const u32x4*>(f + pad_off + u0);
    *reinterpret_cast<u32x4*>(fm + lp * U + u0) = fv;
    if (live) {
      float d[8];
#pragma unroll
      for (int qq = 0; qq < 8; ++qq) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < CM; ++k) a = fmaf(ds[k], wsm[(u0 + qq) * CM + k], a);
        d[qq] = a;
      }
      *reinterpret_cast<u32x4*>(df + pad_off + u0) =
          u32x4{pack_bf16x2(d[0], d[1]), pack_bf16x2(d[2], d[3]), pack_bf16x2(d[4], d[5]), pack_bf16x2(d[6], d[7])};
    }
  }
  __syncthreads();
  // weight gradient: cell (u, k) = sum over this workgroup's pixels of fused[px][u] * dS[px][k], written as this
  // workgroup's slab of partial sums (head_dws_reduce_kernel adds the slabs up: 288 workgroups x 768 atomics onto the
  // same 768 addresses serialised at the memory side and WERE the kernel's run time, 276 of ~300 us)
  for (int cell = threadIdx.x; cell < U * CM; cell += 1024) {
    const int u = cell / CM, k = cell - u * CM;
    float a = 0.f;
    if (k < C) {
#pragma unroll 8
      for (int px = 0; px < 256; ++px) a = fmaf((float)fm[px * U + u], dsm[px * CM + k], a);  // (one chain, in pixel order)
    }
    dws_part[(int64_t)blockIdx.x * U * CM + cell] = a;
  }
}

// dWs[u][k] += sum over the slabs: 16 lanes per cell each take every 16th slab, then a butterfly (fixed order:
// deterministic)
__global__ __launch_bounds__(256) void head_dws_reduce_kernel(const float* __restrict__ part, int nslabs, int U, int CM, int C,
                                                             float* __restrict__ dws) {
  const int cell = blockIdx.x * 16 + (threadIdx.x >> 4), sub = threadIdx.x & 15;
  const bool live = cell < U * CM;
  float a = 0.f;
  if (live)
    for (int b = sub; b < nslabs; b += 16) a += part[(int64_t)b * U * CM + cell];
#pragma unroll
  for (int off = 8; off >= 1; off >>= 1) a += __shfl_xor(a, off, 64);
  if (!live || sub != 0) return;
  const int u = cell / CM, k = cell - u * CM;
  if (k < C) dws[u * C + k] += a;
}

// ---- conv1_1 filter + bias gradient: dW[t][co] = sum_pix in[pix][t] * dy[pix][co], db[co] = sum_pix dy ----
// fp32 input, K = 9*CIN taps.  128-pixel chunks are staged in LDS (input taps fp32, dy bf16); wave w owns 32 of
// the pixels, lane = output channel, all K accumulators in registers: per pixel one 2-byte dy read and K/4
// wave-uniform (broadcast) 16-byte tap reads feed K FMAs.
template <int CIN>
__global__ __launch_bounds__(256) void conv_first_wgrad_kernel(const float* __restrict__ x, const __bf16* __restrict__ dy,
                                                              float* __restrict__ dw, float* __restrict__ db, int N,
                                                              int H, int W, int chunks_per_block,
                                                              float* __restrict__ part_out) {
  constexpr int K = 9 * CIN;
  constexpr int KP = (K + 3) / 4 * 4;  // padded to a multiple of 4 for 16-byte reads
  constexpr int PX = 128;              // pixels per staged chunk
  __shared__ __attribute__((aligned(16))) float ins[PX * KP];
  __shared__ __attribute__((aligned(16))) __bf16 dys[PX * 64];
  const int64_t npix = (int64_t)N * H * W;
  const int co = threadIdx.x & 63, wv = threadIdx.x >> 6;
  typedef __attribute__((ext_vector_type(2))) float f32x2;
  f32x2 acc[KP / 2];  // pairs of taps: one v_pk_fma_f32 per pair and pixel
#pragma unroll
  for (int i = 0; i < KP / 2; ++i) acc[i] = f32x2{0.f, 0.f};
  float bacc = 0.f;
  for (int ch = 0; ch < chunks_per_block; ++ch) {
    const int64_t base = ((int64_t)blockIdx.x * chunks_per_block + ch) * PX;
    if (base >= npix) break;
    __syncthreads();
    // staging: every global load of a thread is issued before its first LDS store (addresses are clamped onto valid
    // pixels and the out-of-image taps selected to zero afterwards, so the loads are unconditional and batch up)
    if (threadIdx.x < PX) {
      const int pix = (int)base + (int)threadIdx.x;  // npix < 2^31 (checked by the launcher): 32-bit divisions
      const bool ok = pix < (int)npix;
      const int pc = ok ? pix : (int)npix - 1;
      const int row = pc / W;
      const int px = pc - row * W, n = row / H, py = row - n * H;
      float v[9][CIN];
      bool inside[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int yy = py + t / 3 - 1, xx = px + t % 3 - 1;
        inside[t] = ok && yy >= 0 && yy < H && xx >= 0 && xx < W;
        const int yc = yy < 0 ? 0 : (yy >= H ? H - 1 : yy), xc = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
        const float* src = x + (((int64_t)n * H + yc) * W + xc) * CIN;
#pragma unroll
        for (int c = 0; c < CIN; ++c) v[t][c] = src[c];
      }
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int c = 0; c < CIN; ++c) ins[threadIdx.x * KP + t * CIN + c] = inside[t] ? v[t][c] : 0.f;
#pragma unroll
      for (int t = K; t < KP; ++t) ins[threadIdx.x * KP + t] = 0.f;
    }
    {
      constexpr int ITERS = PX * 8 / 256;
      u32x4 dv[ITERS];
#pragma unroll
      for (int it = 0; it < ITERS; ++it) {
        const int idx = threadIdx.x + it * 256;
        const int pix = (int)base + (idx >> 3);
        const bool ok = pix < (int)npix;
        const int pc = ok ? pix : (int)npix - 1;
        const int row = pc / W;
        const int px = pc - row * W, n = row / H, py = row - n * H;
        dv[it] = *reinterpret_cast<const u32x4*>(dy + (((int64_t)n * (H + 2) + (py + 1)) * (W + 2) + (px + 1)) * 64 + (idx & 7) * 8);
        if (!ok) dv[it] = u32x4{0u, 0u, 0u, 0u};
      }
#pragma unroll
      for (int it = 0; it < ITERS; ++it) {
        const int idx = threadIdx.x + it * 256;
        *reinterpret_cast<u32x4*>(dys + (idx >> 3) * 64 + (idx & 7) * 8) = dv[it];
      }
    }
    __syncthreads();
#pragma unroll 2
    for (int px = wv * 32; px < wv * 32 + 32; ++px) {
      const float g = (float)dys[px * 64 + co];
      bacc += g;
      const f32x2 g2 = f32x2{g, g};
#pragma unroll
      for (int t4 = 0; t4 < KP; t4 += 4) {
        const f32x4 iv = *reinterpret_cast<const f32x4*>(ins + px * KP + t4);
        acc[t4 / 2] = __builtin_elementwise_fma(f32x2{iv.x, iv.y}, g2, acc[t4 / 2]);
        acc[t4 / 2 + 1] = __builtin_elementwise_fma(f32x2{iv.z, iv.w}, g2, acc[t4 / 2 + 1]);
      }
    }
  }
  if (part_out != nullptr) {
    // deterministic: the four waves (pixel quarters) of a channel take turns adding into LDS, then the workgroup's sums go
    // to its own row of `part_out` ([K + 1][64], row K = the bias) for partials_reduce_kernel
    __syncthreads();
    float* red = ins;  // PX * KP >= (K + 1) * 64 floats
    for (int w = 0; w < 4; ++w) {
      if (wv == w) {
#pragma unroll
        for (int t = 0; t < K; ++t) red[t * 64 + co] = w == 0 ? acc[t / 2][t & 1] : red[t * 64 + co] + acc[t / 2][t & 1];
        red[K * 64 + co] = w == 0 ? bacc : red[K * 64 + co] + bacc;
      }
      __syncthreads();
    }
    for (int c = threadIdx.x; c < (K + 1) * 64; c += 256) part_out[(int64_t)blockIdx.x * (K + 1) * 64 + c] = red[c];
    return;
  }
#pragma unroll
  for (int t = 0; t < K; ++t) atomicAdd(&dw[t * 64 + co], acc[t / 2][t & 1]);
  if (db != nullptr) atomicAdd(&db[co], bacc);
}

// ---- optimizers ([TF1] update rules) ------------------------------------------
// grad_scale folds the 1/world_size of data-parallel averaging into the update.
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                  float* __restrict__ v, int64_t n, float lr_t, float b1, float b2,
                                                  float eps, float gs) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float gi = g[i] * gs;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] -= lr_t * mi / (sqrtf(vi) + eps);
  }
}
__global__ __launch_bounds__(256) void rmsprop_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ ms,
                                                     int64_t n, float lr, float decay, float eps, float gs) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float gi = g[i] * gs;
    const float s = decay * ms[i] + (1.f - decay) * gi * gi;
    ms[i] = s;
    p[i] -= lr * gi / sqrtf(s + eps);
  }
}
__global__ __launch_bounds__(256) void adagrad_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ a,
                                                     int64_t n, float lr, float gs) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float gi = g[i] * gs;
    const float s = a[i] + gi * gi;
    a[i] = s;
    p[i] -= lr * gi / sqrtf(s);
  }
}


// ---- conv1_1 filter + bias gradient on the matrix cores, in exact fp32 -----------------------------------------------
// dW[t][co] = sum_pix X[pix][t] * dY[pix][co] is a (9 CIN + 1) x 64 x pixels GEMM (row 9 CIN: X = 1 gives the bias
// gradient).  v_mfma_f32_16x16x4_f32 takes ONE fp32 element per lane and operand and multiplies exactly like an fmaf chain
// (157 TFLOP/s, the vector rate, but none of the packed-FMA kernel's LDS broadcast reads): lane (i = lane & 15,
// k = lane >> 4) supplies X of pixel p0 + k at tap row 16 tb + i and dY of the same pixel at channels 4 i .. 4 i + 3 (one
// 8-byte load; channel block cb of the MFMA grid is channel 4 i + cb, un-permuted when the sums are written).  A wave
// walks its share of the pixels four quads at a time (the 12 loads of a batch in flight before its 32 MFMAs); the
// workgroup's 8 waves meet in LDS and issue one atomic per cell.
template <int CIN>
__global__ __launch_bounds__(512) void conv_first_wgrad_mfma_kernel(const float* __restrict__ x, const __bf16* __restrict__ dy,
                                                                   float* __restrict__ dw, float* __restrict__ db, int N,
                                                                   int H, int W, int quads_per_wave,
                                                                   float* __restrict__ part_out) {
  constexpr int K = 9 * CIN, TB = (K + 1 + 15) / 16;
  __shared__ float red[TB * 16 * 64];
  for (int c = threadIdx.x; c < TB * 16 * 64; c += 512) red[c] = 0.f;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, kq = lane >> 4;
  // lane constants of the X operand: tap row t = 16 tb + i -> (dy, dx, ci), the bias row, or nothing
  int tdy[TB], tdx[TB], tci[TB], kind[TB];
#pragma unroll
  for (int tb = 0; tb < TB; ++tb) {
    const int t = 16 * tb + i;
    const int tap = t / CIN;
    kind[tb] = t < K ? 0 : (t == K ? 1 : 2);
    tdy[tb] = tap / 3 - 1;
    tdx[tb] = tap % 3 - 1;
    tci[tb] = t - tap * CIN;
  }
  f32x4 acc[TB][4];
#pragma unroll
  for (int tb = 0; tb < TB; ++tb)
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) acc[tb][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
  // A quad = 4 consecutive pixels of one image row (W % 4 == 0, checked by the launcher): its position is wave-uniform
  // and walks incrementally (scalar unit); a lane's tap is at (quad origin) + a lane constant, outside the image only
  // on the quad's edge flags (bit 0 first row, 1 last row, 2 first quad of the row, 3 last quad) -- as in the forward kernel.
  // All loads are bounds-checked BUFFER loads from descriptors that start at the wave's first quad (X: one row and one
  // pixel earlier, so every tap offset is non-negative): a tap outside the image, a row of the operand that is not a tap,
  // a quad past the wave's share get an offset past the descriptor and read 0 -- no branch around any load (the plain
  // loads compiled into an exec-mask branch per load and ~60 scalar instructions per quad of 64-bit addressing: 270 us where
  // the 8 matrix instructions per quad need 125), and the scalar walk is two 32-bit additions.
  constexpr uint32_t OOB = 0x80000000u;
  const int nquads = (N * H * W) >> 2;  // N * H * W < 2^31 (checked by the launcher)
  const int wid = __builtin_amdgcn_readfirstlane(blockIdx.x * 8 + wave);
  const int q0 = wid * quads_per_wave;
  const int q1 = q0 + quads_per_wave < nquads ? q0 + quads_per_wave : nquads;
  const int Wq = W >> 2;
  int qrow = q0 / Wq;                 // n * H + y of the next quad to request
  int qx = (q0 - qrow * Wq) * 4;
  int qn = qrow / H, qy = qrow - qn * H;
  const int back = (W + 1) * CIN * 4;  // bytes the X descriptor starts before the first quad
  uint32_t xoff[TB], kill[TB];
#pragma unroll
  for (int tb = 0; tb < TB; ++tb) {
    xoff[tb] = (uint32_t)(back + ((tdy[tb] * W + tdx[tb] + kq) * CIN + tci[tb]) * 4);
    kill[tb] = 32u | (kind[tb] != 0 ? 16u : ((tdy[tb] < 0 ? 1u : 0u) | (tdy[tb] > 0 ? 2u : 0u) | ((kq == 0 && tdx[tb] < 0) ? 4u : 0u) |
                                             ((kq == 3 && tdx[tb] > 0) ? 8u : 0u)));  // (bit 5: a quad past the wave's share)
  }
  auto wave_rsrc = [](const void* base, int64_t bytes) {  // (built from readfirstlane'd halves: no waterfall loop)
    const uint64_t a = (uint64_t)base;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    const int nrec = __builtin_amdgcn_readfirstlane((int)(bytes < 0x40000000 ? bytes : 0x40000000));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, nrec, 0x00020000);
  };
  // (a wave's share is a few thousand pixels: its offsets stay far below the 1 GB the descriptors are capped at)
  const int64_t xel0 = (int64_t)q0 * 4 * CIN;  // element of the first quad in X (dense [N][H][W][CIN]: linear in the quad)
  const int64_t del0 = (((int64_t)qn * (H + 2) + (qy + 1)) * (W + 2) + (qx + 1)) * 64;
  const auto xrs = wave_rsrc(reinterpret_cast<const char*>(x + xel0) - back, ((int64_t)N * H * W * CIN - xel0) * 4 + back);
  const auto drs = wave_rsrc(dy + del0, ((int64_t)N * (H + 2) * (W + 2) * 64 - del0) * 2);
  const uint32_t dlane = (uint32_t)(kq * 128 + i * 8);
  uint32_t xwalk = 0, dwalk = 0;  // byte offsets of the next quad behind the descriptors' first
  for (int q = q0; q < q1; q += 4) {
    float a[4][TB];
    u32x2 g[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const bool ok = q + b < q1;
      const uint32_t edge = (qy == 0 ? 1u : 0u) | (qy == H - 1 ? 2u : 0u) | (qx == 0 ? 4u : 0u) | (qx == W - 4 ? 8u : 0u) | 16u |
                            (ok ? 0u : 32u);
      // (the out-of-range mark goes into the VECTOR offset: the scalar offset is not part of the descriptor's range check)
      g[b] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(drs, ok ? dlane : OOB, dwalk, 0));
#pragma unroll
      for (int tb = 0; tb < TB; ++tb) {
        const float v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, (kill[tb] & edge) == 0 ? xoff[tb] : OOB,
                                                                                      xwalk, 0));
        a[b][tb] = kind[tb] == 1 ? (ok ? 1.f : 0.f) : v;
      }
      // next quad (wave-uniform walk): X is dense, dY skips its border columns at a row end and a border row pair at an image end
      xwalk += 16 * CIN;
      dwalk += 4 * 128;
      qx += 4;
      if (qx == W) {
        qx = 0;
        qy += 1;
        dwalk += 2 * 128;
        if (qy == H) {
          qy = 0;
          qn += 1;
          dwalk += 2 * (W + 2) * 128;
        }
      }
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const float gf[4] = {bf16_bits_to_f32(g[b].x & 0xffffu), __builtin_bit_cast(float, g[b].x & 0xffff0000u),
                           bf16_bits_to_f32(g[b].y & 0xffffu), __builtin_bit_cast(float, g[b].y & 0xffff0000u)};
#pragma unroll
      for (int tb = 0; tb < TB; ++tb)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) acc[tb][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[b][tb], gf[cb], acc[tb][cb], 0, 0, 0);
    }
  }
  __syncthreads();  // red is zeroed
  // accumulator rows: tap row 16 tb + 4 kq + r, column i of block cb = channel 4 i + cb.  The eight waves add their
  // sums into LDS ONE AFTER THE OTHER (a lane owns its cells within a wave: plain read-modify-write, a fixed order of
  // the eight terms), and the workgroup's sums go to its own row of `part_out` for partials_reduce_kernel: bitwise
  // reproducible.  Without that buffer: atomics onto dw / db as before.
  for (int w = 0; w < 8; ++w) {
    if (wave == w) {
#pragma unroll
      for (int tb = 0; tb < TB; ++tb)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
          for (int r = 0; r < 4; ++r) red[(16 * tb + 4 * kq + r) * 64 + 4 * i + cb] += acc[tb][cb][r];
    }
    __syncthreads();
  }
  for (int c = threadIdx.x; c < (K + 1) * 64; c += 512) {
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

}  // namespace

extern "C" int xv_maxpool2x2_bwd(const xv_act* y, const xv_act* dpooled, const xv_act* dy, void* stream) {
  XV_REQUIRE_BF16(y, dpooled, dy);
  XV_CHECK_ARG(y && dpooled && dy && y->data && dpooled->data && dy->data);
  XV_CHECK_SHAPE(y->c > 0 && (y->c & 7) == 0 && (y->h & 1) == 0 && (y->w & 1) == 0);
  XV_CHECK_SHAPE(dy->n == y->n && dy->h == y->h && dy->w == y->w && dy->c == y->c);
  XV_CHECK_SHAPE(dpooled->n == y->n && dpooled->h == y->h / 2 && dpooled->w == y->w / 2 && dpooled->c == y->c);
  const int64_t total = (int64_t)y->n * (y->h / 2) * (y->w / 2) * (y->c >> 3);
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                     (const __bf16*)y->data, (const __bf16*)dpooled->data, (__bf16*)dy->data, y->n, y->h / 2, y->w / 2, y->c);
  return xv_launch_status();
}

extern "C" int xv_relu_bwd(const xv_act* g, const xv_act* ref, const xv_act* out, void* stream) {
  XV_REQUIRE_BF16(g, ref, out);
  XV_CHECK_ARG(g && ref && out && g->data && ref->data && out->data);
  XV_CHECK_SHAPE(g->n == ref->n && g->h == ref->h && g->w == ref->w && g->c == ref->c && (g->c & 7) == 0);
  XV_CHECK_SHAPE(out->n == g->n && out->h == g->h && out->w == g->w && out->c == g->c);
  const int64_t n8 = (int64_t)g->n * (g->h + 2) * (g->w + 2) * g->c / 8;
  hipLaunchKernelGGL(relu_bwd_kernel, dim3(grid_for(n8)), dim3(256), 0, (hipStream_t)stream, (const __bf16*)g->data,
                     (const __bf16*)ref->data, (__bf16*)out->data, n8);
  return xv_launch_status();
}

extern "C" int xv_upsample2x_bwd(const xv_act* dfused, const xv_act* s5, const xv_act* ds5, void* stream) {
  XV_REQUIRE_BF16(dfused, s5, ds5);
  XV_CHECK_ARG(dfused && s5 && ds5 && dfused->data && s5->data && ds5->data);
  XV_CHECK_SHAPE((s5->c & 7) == 0 && dfused->n == s5->n && dfused->h == 2 * s5->h && dfused->w == 2 * s5->w &&
                 dfused->c == s5->c);
  XV_CHECK_SHAPE(ds5->n == s5->n && ds5->h == s5->h && ds5->w == s5->w && ds5->c == s5->c);
  const int64_t total = (int64_t)s5->n * s5->h * s5->w * (s5->c >> 3);
  hipLaunchKernelGGL(upsample2x_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                     (const __bf16*)dfused->data, (const __bf16*)s5->data, (__bf16*)ds5->data, s5->n, s5->h, s5->w, s5->c);
  return xv_launch_status();
}

extern "C" int xv_count_valid_labels(const int32_t* labels, int num_classes, int64_t npix, int64_t* count, void* stream) {
  XV_CHECK_ARG(labels && count);
  XV_CHECK_SHAPE(npix > 0 && num_classes >= 1);
  // one 1 024-thread workgroup per CU: sixteen waves' loads in flight per CU, and one same-address 64-bit atomic per workgroup
  // (1 024 workgroups of 256 threads: 19 us for the 19 MB label map of 16 images; 256 of 256: 15 us)
  hipLaunchKernelGGL(count_valid_kernel, dim3(grid_for(npix, 1024, 256)), dim3(1024), 0, (hipStream_t)stream, labels,
                     num_classes, npix, reinterpret_cast<unsigned long long*>(count));
  return xv_launch_status();
}

extern "C" size_t xv_decoder_head_bwd_workspace_bytes(int n, int h, int w, int num_classes) {
  if (!xv_dims_sane(n, h, w) || num_classes < 1 || num_classes > 32) return 0;
  const size_t cm = (size_t)(num_classes + 3) / 4 * 4;
  // padded low-resolution scores + the row sums of the score gradient (3 target rows x 8w columns per 1/8-resolution
  // row) + one slab of weight-gradient partial sums per 256 low-resolution pixels (up to 256 decoder units)
  const size_t slabs = ((size_t)n * h * w + 255) / 256;
  // + per workgroup of the loss kernel (at most 4096) its partial bias gradient and loss
  return (((size_t)n * ((size_t)h + 2) * ((size_t)w + 2) + (size_t)n * h * w * 24 + slabs * 256) * cm + 4096 * (cm + 1)) * sizeof(float);
}

extern "C" int xv_score_lowres(const xv_act* fused, const float* w_score, int num_classes, float* S, void* stream);

extern "C" int xv_decoder_head_bwd(const xv_act* fused, const float* w_score, const float* b_score, const int32_t* labels,
                                   const int64_t* valid_count, int num_classes, double* loss, float* dw_score,
                                   float* db_score, const xv_act* dfused, void* workspace, size_t workspace_bytes,
                                   void* stream) {
  XV_REQUIRE_BF16(fused, dfused);
  XV_CHECK_ARG(fused && fused->data && w_score && b_score && labels && valid_count && loss && dw_score && db_score &&
               dfused && dfused->data && workspace);
  XV_CHECK_SHAPE(fused->c > 0 && (fused->c & 7) == 0 && fused->c <= 256 && num_classes >= 1 && num_classes <= 32);
  XV_CHECK_SHAPE(dfused->n == fused->n && dfused->h == fused->h && dfused->w == fused->w && dfused->c == fused->c);
  if (workspace_bytes < xv_decoder_head_bwd_workspace_bytes(fused->n, fused->h, fused->w, num_classes)) return XV_EWORKSPACE;
  XV_CHECK_ARG(((uintptr_t)workspace & 15) == 0);
  hipStream_t s = (hipStream_t)stream;
  const int CM = (num_classes + 3) / 4 * 4;
  float* S = (float*)workspace;
  float* dscore = S + (size_t)fused->n * (fused->h + 2) * (fused->w + 2) * CM;
  int rc = xv_score_lowres(fused, w_score, num_classes, S, stream);
  if (rc != XV_OK) return rc;
  const int64_t npix = (int64_t)fused->n * fused->h * fused->w * 64;
  const int64_t lowres = (int64_t)fused->n * fused->h * fused->w;
  const int64_t ncols = lowres * 8;  // columns of eight output pixels
  XV_CHECK_SHAPE(npix < 0x7fff0000);
  const unsigned g1 = (unsigned)((ncols + 255) / 256 < 4096 ? (ncols + 255) / 256 : 4096), g2 = (unsigned)((lowres + 255) / 256);
  const int U = fused->c;
  float* dws_part = dscore + (size_t)lowres * 24 * CM;  // g2 slabs of U x CM partial sums (U <= 256: sized by the query above)
  float* loss_part = dws_part + (size_t)g2 * U * CM;     // g1 rows of CM + 1 partial sums: bias gradient and loss
  const unsigned long long* cnt = reinterpret_cast<const unsigned long long*>(valid_count);
#define XV_HB(CMV)                                                                                                   \
  {                                                                                                                  \
    if (num_classes == CMV)                                                                                          \
      hipLaunchKernelGGL((head_loss_kernel<CMV, true>), dim3(g1), dim3(256), 0, s, (const float*)S, b_score, labels,   \
                         cnt, fused->n, fused->h, fused->w, num_classes, loss, db_score, dscore, loss_part);          \
    else                                                                                                             \
      hipLaunchKernelGGL(head_loss_kernel<CMV>, dim3(g1), dim3(256), 0, s, (const float*)S, b_score, labels, cnt,     \
                         fused->n, fused->h, fused->w, num_classes, loss, db_score, dscore, loss_part);               \
    hipLaunchKernelGGL(partials_reduce_kernel<256>, dim3(CMV + 1), dim3(256), 0, s, (const float*)loss_part,          \
                       (int)g1, CMV + 1,                                                                              \
                       db_score, num_classes, loss, CMV);                                                             \
    const size_t lds = (size_t)(U * CMV + 256 * CMV) * 4 + (size_t)256 * U * 2;                                       \
    static bool attr[XV_MAX_DEVICES] = {false};                                                                      \
    {                                                                                                                \
      const hipError_t e =                                                                                           \
          xv_allow_dynamic_lds(reinterpret_cast<const void*>(&head_bwd_lowres_kernel<CMV>), 160 * 1024, attr, false);        \
      if (e != hipSuccess) return (int)e;                                                                            \
    }                                                                                                                \
    hipLaunchKernelGGL(head_bwd_lowres_kernel<CMV>, dim3(g2), dim3(1024), lds, s, (const float*)dscore,               \
                       (const __bf16*)fused->data, w_score, fused->n, fused->h, fused->w, U, num_classes, dws_part,   \
                       (__bf16*)dfused->data);                                                                        \
    hipLaunchKernelGGL(head_dws_reduce_kernel, dim3((U * CMV + 15) / 16), dim3(256), 0, s, (const float*)dws_part,    \
                       (int)g2, U, CMV, num_classes, dw_score);                                                       \
  }
  switch (CM / 4) {
    case 1: XV_HB(4) break;
    case 2: XV_HB(8) break;
    case 3: XV_HB(12) break;
    case 4: XV_HB(16) break;
    case 5: XV_HB(20) break;
    case 6: XV_HB(24) break;
    case 7: XV_HB(28) break;
    default: XV_HB(32) break;
  }
#undef XV_HB
  return xv_launch_status();
}

// grid of the first-layer filter-gradient kernels (shared by the launcher and the workspace query)
// conv_wgrad.hip: the bf16-split form (maps that tile in 8x32 pixels); the grid it launches, < 0 where it does not apply
int xv_launch_first_wgrad_split(const float* x, const void* dy, float* dw, float* db, int n, int h, int w, int cin, float* part,
                                int query_only, hipStream_t stream);
static bool first_wgrad_split_ok() {
  static const bool on = getenv("XV_FIRST_WGRAD_OLD") == nullptr &&
                         (getenv("XV_FIRST_WGRAD_SPLIT") == nullptr || atoi(getenv("XV_FIRST_WGRAD_SPLIT")) != 0);
  return on;
}

static void first_wgrad_geometry(int64_t npix, int w, int cin, bool& mfma, unsigned& grid, int& per) {
  static const bool use_old = getenv("XV_FIRST_WGRAD_OLD") != nullptr;  // the packed-FMA kernel (A/B timing)
  mfma = !use_old && (w & 3) == 0 && npix * cin < 0x7ff00000;
  if (mfma) {
    // 4 workgroups of 8 waves per CU (a bounded grid: each ends with (9 cin + 1) * 64 partial sums); 16 images at
    // 768x384: 317 us RGB / 181 us depth (packed-FMA kernel: 537 / 340); widths that are not a multiple of 4 keep that kernel
    const int64_t nquads = (npix + 3) / 4;
    static const int per_cu = getenv("XV_FIRST_WGRAD_PER_CU") ? atoi(getenv("XV_FIRST_WGRAD_PER_CU")) : 4;
    int64_t blocks = (int64_t)xv_num_cus() * (per_cu > 0 ? per_cu : 4);
    if (blocks * 8 * 4 > nquads) blocks = (nquads + 31) / 32;
    int64_t qpw = (nquads + blocks * 8 - 1) / (blocks * 8);
    qpw = (qpw + 3) / 4 * 4;
    grid = (unsigned)((nquads + qpw * 8 - 1) / (qpw * 8));
    per = (int)qpw;
    return;
  }
  // at most 512 workgroups (see xv_score_dense_bwd)
  int chunks = (int)((npix + 128 * 512 - 1) / (128 * 512));
  if (chunks < 32) chunks = 32;
  grid = (unsigned)((npix + 128 * (int64_t)chunks - 1) / (128 * (int64_t)chunks));
  per = chunks;
}

extern "C" size_t xv_conv2d_first_bwd_filter_workspace_bytes(int n, int h, int w, int cin) {
  if (!xv_dims_sane(n, h, w) || cin < 1 || cin > 4 || (int64_t)n * h * w >= 0x7fff0000) return 0;
  bool mfma;
  unsigned grid;
  int per;
  first_wgrad_geometry((int64_t)n * h * w, w, cin, mfma, grid, per);
  const int gs = first_wgrad_split_ok() ? xv_launch_first_wgrad_split(nullptr, nullptr, nullptr, nullptr, n, h, w, cin, nullptr, 1, nullptr) : -1;
  if (gs > (int)grid) grid = (unsigned)gs;
  return (size_t)grid * (9 * cin + 1) * 64 * sizeof(float);
}

extern "C" int xv_conv2d_first_bwd_filter_ws(const float* x, int n, int h, int w, int cin, const xv_act* dy, float* dw_hwio,
                                             float* dbias, void* workspace, size_t workspace_bytes, void* stream) {
  XV_REQUIRE_BF16(dy);
  XV_CHECK_ARG(x && dy && dy->data && dw_hwio);
  XV_CHECK_SHAPE(n > 0 && h > 0 && w > 0 && cin >= 1 && cin <= 4 && dy->n == n && dy->h == h && dy->w == w && dy->c == 64);
  XV_CHECK_SHAPE((int64_t)n * h * w < 0x7fff0000);
  const int64_t npix = (int64_t)n * h * w;
  hipStream_t s = (hipStream_t)stream;
  const __bf16* g = (const __bf16*)dy->data;
  bool mfma;
  unsigned grid;
  int per;
  first_wgrad_geometry(npix, w, cin, mfma, grid, per);
  // with a workspace: per-workgroup partial sums + a fixed-order reduce (bitwise reproducible); without: atomics
  float* part = nullptr;
  if (workspace != nullptr) {
    if (workspace_bytes < xv_conv2d_first_bwd_filter_workspace_bytes(n, h, w, cin)) return XV_EWORKSPACE;
    XV_CHECK_ARG(((uintptr_t)workspace & 15) == 0 && dbias != nullptr);
    part = (float*)workspace;
  }
  // the bf16-split form on maps that tile in 8x32 pixels (XV_FIRST_WGRAD_SPLIT=0: the fp32 matrix instruction everywhere)
  const int gs = first_wgrad_split_ok() ? xv_launch_first_wgrad_split(x, g, dw_hwio, dbias, n, h, w, cin, part, 0, s) : -1;
  if (gs == -2) return XV_EINVAL;
  if (gs > 0) {
    grid = (unsigned)gs;
  } else if (mfma) {
    switch (cin) {
      case 1: hipLaunchKernelGGL(conv_first_wgrad_mfma_kernel<1>, dim3(grid), dim3(512), 0, s, x, g, dw_hwio, dbias, n, h, w, per, part); break;
      case 2: hipLaunchKernelGGL(conv_first_wgrad_mfma_kernel<2>, dim3(grid), dim3(512), 0, s, x, g, dw_hwio, dbias, n, h, w, per, part); break;
      case 3: hipLaunchKernelGGL(conv_first_wgrad_mfma_kernel<3>, dim3(grid), dim3(512), 0, s, x, g, dw_hwio, dbias, n, h, w, per, part); break;
      default: hipLaunchKernelGGL(conv_first_wgrad_mfma_kernel<4>, dim3(grid), dim3(512), 0, s, x, g, dw_hwio, dbias, n, h, w, per, part); break;
    }
  } else {
    switch (cin) {
      case 1: hipLaunchKernelGGL(conv_first_wgrad_kernel<1>, dim3(grid), dim3(256), 0, s, x, g, dw_hwio, dbias, n, h, w, per, part); break;
      case 2: hipLaunchKernelGGL(conv_first_wgrad_kernel<2>, dim3(grid), dim3(256), 0, s, x, g, dw_hwio, dbias, n, h, w, per, part); break;
      case 3: hipLaunchKernelGGL(conv_first_wgrad_kernel<3>, dim3(grid), dim3(256), 0, s, x, g, dw_hwio, dbias, n, h, w, per, part); break;
      default: hipLaunchKernelGGL(conv_first_wgrad_kernel<4>, dim3(grid), dim3(256), 0, s, x, g, dw_hwio, dbias, n, h, w, per, part); break;
    }
  }
  if (part != nullptr) {
    // dW [9 cin][64] and db [64] are contiguous rows of a workgroup's partial block: two reduce launches into the two outputs
    const int len = (9 * cin + 1) * 64;
    hipLaunchKernelGGL(partials_reduce_kernel<64>, dim3((9 * cin * 64 + 3) / 4), dim3(256), 0, s, (const float*)part, (int)grid,
                       len, dw_hwio, 9 * cin * 64, (double*)nullptr, -1);
    hipLaunchKernelGGL(partials_reduce_kernel<256>, dim3(64), dim3(256), 0, s, (const float*)part + 9 * cin * 64, (int)grid, len,
                       dbias, 64, (double*)nullptr, -1);
  }
  return xv_launch_status();
}

extern "C" int xv_conv2d_first_bwd_filter(const float* x, int n, int h, int w, int cin, const xv_act* dy, float* dw_hwio,
                                          float* dbias, void* stream) {
  return xv_conv2d_first_bwd_filter_ws(x, n, h, w, cin, dy, dw_hwio, dbias, nullptr, 0, stream);
}

extern "C" int xv_adam_step(float* param, const float* grad, float* m, float* v, int64_t n, float lr_t, float beta1,
                            float beta2, float eps, float grad_scale, void* stream) {
  XV_CHECK_ARG(param && grad && m && v);
  XV_CHECK_SHAPE(n > 0);
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, param, grad, m, v, n, lr_t, beta1,
                     beta2, eps, grad_scale);
  return xv_launch_status();
}

extern "C" int xv_rmsprop_step(float* param, const float* grad, float* ms, int64_t n, float lr, float decay, float eps,
                               float grad_scale, void* stream) {
  XV_CHECK_ARG(param && grad && ms);
  XV_CHECK_SHAPE(n > 0);
  hipLaunchKernelGGL(rmsprop_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, param, grad, ms, n, lr, decay,
                     eps, grad_scale);
  return xv_launch_status();
}

extern "C" int xv_adagrad_step(float* param, const float* grad, float* accum, int64_t n, float lr, float grad_scale,
                               void* stream) {
  XV_CHECK_ARG(param && grad && accum);
  XV_CHECK_SHAPE(n > 0);
  hipLaunchKernelGGL(adagrad_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, param, grad, accum, n, lr,
                     grad_scale);
  return xv_launch_status();
}
