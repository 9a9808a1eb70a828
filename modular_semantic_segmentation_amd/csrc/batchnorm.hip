// Training-mode batch normalisation and the un-commuted training head for gfx950.
//
// Replaces tf.layers.batch_normalization(training=True) between a conv / deconv and its activation
// (custom_layers.py:112-119,124-139) and its gradient, as the reference trains SimpleFCN with
// `batch_normalization: true` (experiments/example_config.yaml).  [TF1] semantics: statistics over (N, H, W) per
// channel, normalisation with the BIASED batch variance, epsilon 1e-3, moving averages with momentum 0.99 fed the
// UNBIASED variance (fused kernel), gamma / beta trainable.
//
// With a batch norm between the x8 deconv and its relu the decoder head no longer commutes (see
// decoder_head_affine_kernel), so training materialises the full-resolution U-channel map: raw bilinear up-sampling
// (forward and transpose), a per-pixel U->C score conv and a dense softmax cross-entropy.  All HBM-bound.
#include "xv_common.h"

namespace {

__device__ __forceinline__ float bf_lo(uint32_t v) { return __builtin_bit_cast(float, v << 16); }
__device__ __forceinline__ float bf_hi(uint32_t v) { return __builtin_bit_cast(float, v & 0xffff0000u); }

inline int bn_grid(int64_t total, int cap = 2048) {
  int64_t g = (total + 255) / 256;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

// ---- per-channel sums over the interior of a padded-NHWC bf16 tensor ----------------------------------------
// weight of source pixel i in output pixel o of the bilinear x S transposed conv (0 if not a tap; custom_layers.py:8-25)
template <int S>
__device__ __forceinline__ float bl_w(int o, int i) {
  constexpr float center = (2.f * S - 1.f - (S % 2)) / (2.f * S);
  const int p = o + S / 2 - i * S;  // tap index along the kernel: 0 .. 2S-1
  if (p < 0 || p >= 2 * S) return 0.f;
  return 1.f - fabsf((float)p / S - center);
}
// Eight channels (cg) of output pixel (oy, ox) of the x S up-sampled map of the padded low-resolution map x, rounded to
// bf16: what upsample_raw_kernel stores -- an explicit fmaf chain, which the batch-norm passes of the layer behind the x8 deconv
// (bn_ups8_*_kernel below) repeat with compile-time column weights instead of reading the 0.6 GB map back: same bits.
template <int S>
__device__ __forceinline__ u32x4 upsample_words(const __bf16* __restrict__ x, int n, int oy, int ox, int cg, int Hi, int Wi, int C) {
  const int iy1 = (oy + S / 2) / S, ix1 = (ox + S / 2) / S;  // taps iy1-1, iy1 (padded coords iy1, iy1+1)
  float wy1, wy0, wx1, wx0;
  if constexpr (S == 8) {
    // bl_w in closed form: with tap index p = (o + 4) & 7 of the nearer source, w1 = (2 p + 1) / 16 and w0 = (15 - 2 p) / 16 --
    // the same floats (every term of bl_w's expression is a dyadic rational of a few bits: exact), without its branch,
    // division and fabs (recomputing the map was ALU-bound: 230 us a pass)
    const int py = (oy + 4) & 7, px = (ox + 4) & 7;
    wy1 = (float)(2 * py + 1) * 0.0625f, wy0 = (float)(15 - 2 * py) * 0.0625f;
    wx1 = (float)(2 * px + 1) * 0.0625f, wx0 = (float)(15 - 2 * px) * 0.0625f;
  } else {
    wy1 = bl_w<S>(oy, iy1), wy0 = bl_w<S>(oy, iy1 - 1), wx1 = bl_w<S>(ox, ix1), wx0 = bl_w<S>(ox, ix1 - 1);
  }
  const __bf16* p00 = x + (((int64_t)n * (Hi + 2) + iy1) * (Wi + 2) + ix1) * C + cg * 8;
  const int64_t rowp = (int64_t)(Wi + 2) * C;
  const u32x4 a00 = *reinterpret_cast<const u32x4*>(p00), a01 = *reinterpret_cast<const u32x4*>(p00 + C);
  const u32x4 a10 = *reinterpret_cast<const u32x4*>(p00 + rowp), a11 = *reinterpret_cast<const u32x4*>(p00 + rowp + C);
  const float w00 = wy0 * wx0, w01 = wy0 * wx1, w10 = wy1 * wx0, w11 = wy1 * wx1;
  u32x4 o;
#pragma unroll
  for (int w = 0; w < 4; ++w)
    o[w] = pack_bf16x2(fmaf(bf_lo(a11[w]), w11, fmaf(bf_lo(a10[w]), w10, fmaf(bf_lo(a01[w]), w01, bf_lo(a00[w]) * w00))),
                       fmaf(bf_hi(a11[w]), w11, fmaf(bf_hi(a10[w]), w10, fmaf(bf_hi(a01[w]), w01, bf_hi(a00[w]) * w00))));
  return o;
}

// MODE 0: sums[c] = sum z, sums[C + c] = sum z^2                                     (forward statistics)
// MODE 1: sums[c] = sum g, sums[C + c] = sum g * zhat, g = dy * (y > 0 or 1)        (backward reductions)
// MODE 2: as 1 with the relu mask RECOMPUTED from z (z * scale + shift > 0, the forward pass's own expression) instead
//         of read from the activation map: a third less traffic (`y` then carries scale, `zsh` shift)
// A thread owns one 8-channel group (the grid stride is a multiple of C/8), accumulates in fp32 over its pixels,
// the block reduces through LDS and issues one double atomic per channel.
template <int MODE>
__global__ __launch_bounds__(256) void bn_reduce_kernel(const __bf16* __restrict__ z, const __bf16* __restrict__ dy,
                                                       const __bf16* __restrict__ y, const float* __restrict__ mean,
                                                       const float* __restrict__ invstd, double* __restrict__ sums,
                                                       int N, int H, int W, int C, const float* __restrict__ zsc = nullptr,
                                                       const float* __restrict__ zsh = nullptr, float* __restrict__ part = nullptr) {
  const int c8 = C >> 3;
  const int64_t total = (int64_t)N * H * W * c8;
  const int cg = threadIdx.x % c8;  // constant over the loop: 256 and gridDim.x * 256 are multiples of c8
  float s0[8], s1[8], mu[8], is[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    s0[e] = s1[e] = 0.f;
    mu[e] = MODE >= 1 ? mean[cg * 8 + e] : 0.f;
    is[e] = MODE >= 1 ? invstd[cg * 8 + e] : 0.f;
  }
  float sc[8], sh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) sc[e] = MODE == 2 ? zsc[cg * 8 + e] : 0.f, sh[e] = MODE == 2 ? zsh[cg * 8 + e] : 0.f;
  // total < 2^31 (checked by the launchers): 32-bit index arithmetic
  const int total32 = (int)total, stride = (int)gridDim.x * 256;
  // The positions idx0, idx0 + stride, idx0 + 2 stride, ... are visited IN ORDER, and the stride is a whole number of pixels
  // (a multiple of c8): (n, y, x) of a position follow from the previous one by constant increments with two carries --
  // the three 32-bit divisions per 16-byte element of the first form were more instructions than the rest of the pass
  // (bn_reduce<0> on a 0.6 GB map: 165 us, ALU-bound).
  int wk_x, wk_y, wk_n;
  {
    const int p0 = ((int)blockIdx.x * 256 + (int)threadIdx.x) / c8, row0 = p0 / W;
    wk_x = p0 - row0 * W, wk_n = row0 / H, wk_y = row0 - wk_n * H;
  }
  const int wk_dp = stride / c8, wk_drow = wk_dp / W, wk_dx = wk_dp - wk_drow * W, wk_dn = wk_drow / H, wk_dy = wk_drow - wk_dn * H;
  auto next_offset = [&]() -> int64_t {
    const int64_t off = (((int64_t)wk_n * (H + 2) + wk_y + 1) * (W + 2) + wk_x + 1) * C + cg * 8;
    wk_x += wk_dx;
    const int cx = wk_x >= W ? 1 : 0;
    wk_x -= cx ? W : 0;
    wk_y += wk_dy + cx;
    const int cy = wk_y >= H ? 1 : 0;
    wk_y -= cy ? H : 0;
    wk_n += wk_dn + cy;
    return off;
  };
  int idx = (int)blockIdx.x * 256 + (int)threadIdx.x;
  if (MODE == 0) {
    // a pure streaming read: four independent 16-byte loads in flight per thread (with one, the 1024-workgroup grid
    // keeps only 4 MB in flight, a third of what HBM latency x bandwidth asks for, and the kernel ran at 2.1 TB/s)
    for (; idx + 3 * stride < total32; idx += 4 * stride) {
      u32x4 v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const u32x4*>(z + next_offset());
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const float a = bf_lo(v[q][w]), b = bf_hi(v[q][w]);
          s0[2 * w] += a;
          s1[2 * w] += a * a;
          s0[2 * w + 1] += b;
          s1[2 * w + 1] += b * b;
        }
    }
  }
  if (MODE == 2) {
    for (; idx + stride < total32; idx += 2 * stride) {
      u32x4 zv[2], gv[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int64_t off = next_offset();
        zv[q] = *reinterpret_cast<const u32x4*>(z + off);
        gv[q] = *reinterpret_cast<const u32x4*>(dy + off);
      }
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const float za = bf_lo(zv[q][w]), zb = bf_hi(zv[q][w]);
          const float ga = fmaf(za, sc[2 * w], sh[2 * w]) > 0.f ? bf_lo(gv[q][w]) : 0.f;
          const float gb = fmaf(zb, sc[2 * w + 1], sh[2 * w + 1]) > 0.f ? bf_hi(gv[q][w]) : 0.f;
          s0[2 * w] += ga;
          s1[2 * w] += ga * (za - mu[2 * w]) * is[2 * w];
          s0[2 * w + 1] += gb;
          s1[2 * w + 1] += gb * (zb - mu[2 * w + 1]) * is[2 * w + 1];
        }
    }
    for (; idx < total32; idx += stride) {
      const int64_t off = next_offset();
      const u32x4 zv = *reinterpret_cast<const u32x4*>(z + off), gv = *reinterpret_cast<const u32x4*>(dy + off);
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const float za = bf_lo(zv[w]), zb = bf_hi(zv[w]);
        const float ga = fmaf(za, sc[2 * w], sh[2 * w]) > 0.f ? bf_lo(gv[w]) : 0.f;
        const float gb = fmaf(zb, sc[2 * w + 1], sh[2 * w + 1]) > 0.f ? bf_hi(gv[w]) : 0.f;
        s0[2 * w] += ga;
        s1[2 * w] += ga * (za - mu[2 * w]) * is[2 * w];
        s0[2 * w + 1] += gb;
        s1[2 * w + 1] += gb * (zb - mu[2 * w + 1]) * is[2 * w + 1];
      }
    }
  }
  if (MODE == 1) {
    // three streams (z, dy, y): two pixels' loads in flight per thread, accumulated in pixel order (the sums do not depend
    // on the unrolling)
    const u32x4 ones = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};  // no relu mask
    for (; idx + stride < total32; idx += 2 * stride) {
      u32x4 zv[2], gv[2], yv[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int64_t off = next_offset();
        zv[q] = *reinterpret_cast<const u32x4*>(z + off);
        gv[q] = *reinterpret_cast<const u32x4*>(dy + off);
        yv[q] = y ? *reinterpret_cast<const u32x4*>(y + off) : ones;
      }
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const float ga = bf_lo(yv[q][w]) > 0.f ? bf_lo(gv[q][w]) : 0.f, gb = bf_hi(yv[q][w]) > 0.f ? bf_hi(gv[q][w]) : 0.f;
          s0[2 * w] += ga;
          s1[2 * w] += ga * (bf_lo(zv[q][w]) - mu[2 * w]) * is[2 * w];
          s0[2 * w + 1] += gb;
          s1[2 * w + 1] += gb * (bf_hi(zv[q][w]) - mu[2 * w + 1]) * is[2 * w + 1];
        }
    }
  }
  for (; idx < total32; idx += stride) {
    const int64_t off = next_offset();
    const u32x4 zv = *reinterpret_cast<const u32x4*>(z + off);
    if (MODE == 0) {
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const float a = bf_lo(zv[w]), b = bf_hi(zv[w]);
        s0[2 * w] += a;
        s1[2 * w] += a * a;
        s0[2 * w + 1] += b;
        s1[2 * w + 1] += b * b;
      }
    } else {
      const u32x4 gv = *reinterpret_cast<const u32x4*>(dy + off);
      u32x4 yv = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};  // ones: no relu mask
      if (y) yv = *reinterpret_cast<const u32x4*>(y + off);
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const float ga = bf_lo(yv[w]) > 0.f ? bf_lo(gv[w]) : 0.f, gb = bf_hi(yv[w]) > 0.f ? bf_hi(gv[w]) : 0.f;
        s0[2 * w] += ga;
        s1[2 * w] += ga * (bf_lo(zv[w]) - mu[2 * w]) * is[2 * w];
        s0[2 * w + 1] += gb;
        s1[2 * w + 1] += gb * (bf_hi(zv[w]) - mu[2 * w + 1]) * is[2 * w + 1];
      }
    }
  }
  __shared__ float red[256][17];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    red[threadIdx.x][e] = s0[e];
    red[threadIdx.x][8 + e] = s1[e];
  }
  __syncthreads();
  // thread t < c8 * 16 sums column (t % 16) of the rows with cg == t / 16
  for (int t = threadIdx.x; t < c8 * 16; t += 256) {
    const int g = t >> 4, col = t & 15;
    float acc = 0.f;
    for (int row = g; row < 256; row += c8) acc += red[row][col];
    const int ch = g * 8 + (col & 7);
    // with a workspace the workgroup's sums go to its own row, added up in workgroup order by bn_sums_kernel (bitwise
    // reproducible statistics); without one: f64 atomics (order not fixed, differences at 1e-16 relative)
    if (part != nullptr)
      part[(int64_t)blockIdx.x * 2 * C + (col < 8 ? ch : C + ch)] = acc;
    else
      atomicAdd(sums + (col < 8 ? ch : C + ch), (double)acc);
  }
}

// sums[i] = sum over the workgroups' partial rows, in a fixed tree (one workgroup per element: lane l adds rows l, l + 256,
// ... in order, a butterfly inside each wave, the four waves in wave order): overwrites `sums` (no memset needed)
// dgamma / dbeta (backward reductions, len = 2 C: sums[c] = sum dy, sums[C + c] = sum dy xhat): the gradient accumulation of
// bn_grads_kernel in the same launch -- dbeta[c] += sums[c], dgamma[c] += sums[C + c] (18 five-microsecond launches per
// training step with batch norm)
__global__ __launch_bounds__(256) void bn_sums_kernel(const float* __restrict__ part, int nblocks, int len,
                                                     double* __restrict__ sums, float* __restrict__ dgamma = nullptr,
                                                     float* __restrict__ dbeta = nullptr) {
  __shared__ double wsum[4];
  const int i = blockIdx.x;
  double a = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += 256) a += (double)part[(int64_t)b * len + i];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) a += __shfl_xor(a, off, 64);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double t = ((wsum[0] + wsum[1]) + wsum[2]) + wsum[3];
    sums[i] = t;
    if (dgamma != nullptr) {
      const int C = len >> 1;
      if (i < C)
        dbeta[i] += (float)t;
      else
        dgamma[i - C] += (float)t;
    }
  }
}

// mean / biased variance -> invstd, scale = gamma * invstd, shift = beta - mean * scale; moving statistics
__global__ void bn_finalize_kernel(const double* __restrict__ sums, int C, double M, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float eps, float momentum,
                                   float* __restrict__ moving_mean, float* __restrict__ moving_var,
                                   float* __restrict__ mean, float* __restrict__ invstd, float* __restrict__ scale,
                                   float* __restrict__ shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double mu = sums[c] / M;
  double var = sums[C + c] / M - mu * mu;
  var = var > 0.0 ? var : 0.0;
  const float is = (float)(1.0 / sqrt(var + (double)eps));
  mean[c] = (float)mu;
  invstd[c] = is;
  scale[c] = gamma[c] * is;
  shift[c] = beta[c] - (float)mu * gamma[c] * is;
  if (moving_mean) {
    const double unbiased = M > 1.0 ? var * M / (M - 1.0) : var;
    moving_mean[c] = moving_mean[c] * momentum + (float)mu * (1.f - momentum);
    moving_var[c] = moving_var[c] * momentum + (float)unbiased * (1.f - momentum);
  }
}

// bn_sums_kernel for the two columns of one channel (the same tree per column: the same bits in `sums`) + bn_finalize_kernel
// for that channel, in ONE launch: one workgroup per channel (18 five-microsecond launches per training step with batch norm)
__global__ __launch_bounds__(256) void bn_sums_finalize_kernel(const float* __restrict__ part, int nblocks, int C,
                                                              double* __restrict__ sums, double M,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              float eps, float momentum, float* __restrict__ moving_mean,
                                                              float* __restrict__ moving_var, float* __restrict__ mean,
                                                              float* __restrict__ invstd, float* __restrict__ scale,
                                                              float* __restrict__ shift) {
  __shared__ double wsum[2][4];
  const int c = blockIdx.x;
  double a0 = 0.0, a1 = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += 256) {
    a0 += (double)part[(int64_t)b * 2 * C + c];
    a1 += (double)part[(int64_t)b * 2 * C + C + c];
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    a0 += __shfl_xor(a0, off, 64);
    a1 += __shfl_xor(a1, off, 64);
  }
  if ((threadIdx.x & 63) == 0) wsum[0][threadIdx.x >> 6] = a0, wsum[1][threadIdx.x >> 6] = a1;
  __syncthreads();
  if (threadIdx.x != 0) return;
  const double s0 = ((wsum[0][0] + wsum[0][1]) + wsum[0][2]) + wsum[0][3];
  const double s1 = ((wsum[1][0] + wsum[1][1]) + wsum[1][2]) + wsum[1][3];
  sums[c] = s0;
  sums[C + c] = s1;
  const double mu = s0 / M;
  double var = s1 / M - mu * mu;
  var = var > 0.0 ? var : 0.0;
  const float is = (float)(1.0 / sqrt(var + (double)eps));
  mean[c] = (float)mu;
  invstd[c] = is;
  scale[c] = gamma[c] * is;
  shift[c] = beta[c] - (float)mu * gamma[c] * is;
  if (moving_mean) {
    const double unbiased = M > 1.0 ? var * M / (M - 1.0) : var;
    moving_mean[c] = moving_mean[c] * momentum + (float)mu * (1.f - momentum);
    moving_var[c] = moving_var[c] * momentum + (float)unbiased * (1.f - momentum);
  }
}

// y = [relu](z * scale + shift) over the interior
__global__ __launch_bounds__(256) void bn_apply_kernel(const __bf16* __restrict__ z, const float* __restrict__ scale,
                                                      const float* __restrict__ shift, int relu,
                                                      __bf16* __restrict__ y, int N, int H, int W, int C) {
  const int c8 = C >> 3;
  const int64_t total = (int64_t)N * H * W * c8;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int cg = (int)(idx % c8);
    int64_t r = idx / c8;
    const int x = (int)(r % W);
    r /= W;
    const int yy = (int)(r % H);
    const int n = (int)(r / H);
    const int64_t off = (((int64_t)n * (H + 2) + yy + 1) * (W + 2) + x + 1) * C + cg * 8;
    const u32x4 zv = *reinterpret_cast<const u32x4*>(z + off);
    u32x4 o;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      float a = fmaf(bf_lo(zv[w]), scale[cg * 8 + 2 * w], shift[cg * 8 + 2 * w]);
      float b = fmaf(bf_hi(zv[w]), scale[cg * 8 + 2 * w + 1], shift[cg * 8 + 2 * w + 1]);
      if (relu) {
        a = a > 0.f ? a : 0.f;
        b = b > 0.f ? b : 0.f;
      }
      o[w] = pack_bf16x2(a, b);
    }
    *reinterpret_cast<u32x4*>(y + off) = o;
  }
}

// dz = scale * (g - dbeta / M - zhat * dgamma / M), g = dy * (y > 0 or 1)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const __bf16* __restrict__ dy, const __bf16* __restrict__ y,
                                                          const __bf16* __restrict__ z, const float* __restrict__ mean,
                                                          const float* __restrict__ invstd,
                                                          const float* __restrict__ gamma,
                                                          const double* __restrict__ sums, double M,
                                                          __bf16* __restrict__ dz, int N, int H, int W, int C) {
  const int c8 = C >> 3;
  const int64_t total = (int64_t)N * H * W * c8;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int cg = (int)(idx % c8);
    int64_t r = idx / c8;
    const int x = (int)(r % W);
    r /= W;
    const int yy = (int)(r % H);
    const int n = (int)(r / H);
    const int64_t off = (((int64_t)n * (H + 2) + yy + 1) * (W + 2) + x + 1) * C + cg * 8;
    const u32x4 zv = *reinterpret_cast<const u32x4*>(z + off);
    const u32x4 gv = *reinterpret_cast<const u32x4*>(dy + off);
    u32x4 yv = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    if (y) yv = *reinterpret_cast<const u32x4*>(y + off);
    u32x4 o;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      float v[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int c = cg * 8 + 2 * w + h;
        const float yy_ = h ? bf_hi(yv[w]) : bf_lo(yv[w]);
        const float g = yy_ > 0.f ? (h ? bf_hi(gv[w]) : bf_lo(gv[w])) : 0.f;
        const float zh = ((h ? bf_hi(zv[w]) : bf_lo(zv[w])) - mean[c]) * invstd[c];
        const float db = (float)(sums[c] / M), dg = (float)(sums[C + c] / M);
        v[h] = gamma[c] * invstd[c] * (g - db - zh * dg);
      }
      o[w] = pack_bf16x2(v[0], v[1]);
    }
    *reinterpret_cast<u32x4*>(dz + off) = o;
  }
}

// ---- the same two passes for the channel counts of the trunk (C / 8 divides 256: 64 .. 2048 channels) --------------
// A thread keeps ONE 8-channel group for the whole launch (the grid stride is a multiple of C / 8), so the per-channel
// constants are loaded -- and the two double divisions per channel of the backward pass done -- once per thread instead of
// once per element (the generic kernels above ran at 1.9 TB/s on them); 32-bit index arithmetic; several pixels' loads in
// flight per thread.  Same expressions, same bits.
__global__ __launch_bounds__(256) void bn_apply_fast_kernel(const __bf16* __restrict__ z, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, int relu,
                                                           __bf16* __restrict__ y, int N, int H, int W, int C) {
  const int c8 = C >> 3;
  const int cg = threadIdx.x % c8;
  float sc[8], sh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) sc[e] = scale[cg * 8 + e], sh[e] = shift[cg * 8 + e];
  const int total = N * H * W * c8, stride = (int)gridDim.x * 256;  // < 2^31 (checked by the launcher)
  // The positions idx0, idx0 + stride, idx0 + 2 stride, ... are visited IN ORDER, and the stride is a whole number of pixels
  // (a multiple of c8): (n, y, x) of a position follow from the previous one by constant increments with two carries --
  // the three 32-bit divisions per 16-byte element of the first form were more instructions than the rest of the pass
  // (bn_reduce<0> on a 0.6 GB map: 165 us, ALU-bound).
  int wk_x, wk_y, wk_n;
  {
    const int p0 = ((int)blockIdx.x * 256 + (int)threadIdx.x) / c8, row0 = p0 / W;
    wk_x = p0 - row0 * W, wk_n = row0 / H, wk_y = row0 - wk_n * H;
  }
  const int wk_dp = stride / c8, wk_drow = wk_dp / W, wk_dx = wk_dp - wk_drow * W, wk_dn = wk_drow / H, wk_dy = wk_drow - wk_dn * H;
  auto next_offset = [&]() -> int64_t {
    const int64_t off = (((int64_t)wk_n * (H + 2) + wk_y + 1) * (W + 2) + wk_x + 1) * C + cg * 8;
    wk_x += wk_dx;
    const int cx = wk_x >= W ? 1 : 0;
    wk_x -= cx ? W : 0;
    wk_y += wk_dy + cx;
    const int cy = wk_y >= H ? 1 : 0;
    wk_y -= cy ? H : 0;
    wk_n += wk_dn + cy;
    return off;
  };
  auto apply = [&](const u32x4 zv) {
    u32x4 o;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      float a = fmaf(bf_lo(zv[w]), sc[2 * w], sh[2 * w]);
      float b = fmaf(bf_hi(zv[w]), sc[2 * w + 1], sh[2 * w + 1]);
      if (relu) {
        a = a > 0.f ? a : 0.f;
        b = b > 0.f ? b : 0.f;
      }
      o[w] = pack_bf16x2(a, b);
    }
    return o;
  };
  int idx = (int)blockIdx.x * 256 + (int)threadIdx.x;
  for (; idx + 3 * stride < total; idx += 4 * stride) {
    int64_t off[4];
    u32x4 v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      off[q] = next_offset();
      v[q] = *reinterpret_cast<const u32x4*>(z + off[q]);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) *reinterpret_cast<u32x4*>(y + off[q]) = apply(v[q]);
  }
  for (; idx < total; idx += stride) {
    const int64_t off = next_offset();
    *reinterpret_cast<u32x4*>(y + off) = apply(*reinterpret_cast<const u32x4*>(z + off));
  }
}

// MASKZ: the relu mask recomputed from z (fmaf(z, zsc, zsh) > 0: the SAME explicit fused multiply-add in the forward apply kernels and in every
// kernel that recomputes the mask, so the sign test is identical by construction, not by the compiler's contraction choices) instead of read from y
template <bool MASKZ>
__global__ __launch_bounds__(256) void bn_bwd_apply_fast_kernel(const __bf16* __restrict__ dy, const __bf16* __restrict__ y,
                                                               const __bf16* __restrict__ z, const float* __restrict__ mean,
                                                               const float* __restrict__ invstd,
                                                               const float* __restrict__ gamma,
                                                               const double* __restrict__ sums, double M,
                                                               __bf16* __restrict__ dz, int N, int H, int W, int C,
                                                               const float* __restrict__ zsc, const float* __restrict__ zsh) {
  const int c8 = C >> 3;
  const int cg = threadIdx.x % c8;
  float mu[8], is[8], gk[8], db[8], dg[8], sc[8], sh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = cg * 8 + e;
    mu[e] = mean[c], is[e] = invstd[c], gk[e] = gamma[c];
    db[e] = (float)(sums[c] / M), dg[e] = (float)(sums[C + c] / M);
    sc[e] = MASKZ ? zsc[c] : 0.f, sh[e] = MASKZ ? zsh[c] : 0.f;
  }
  const int total = N * H * W * c8, stride = (int)gridDim.x * 256;
  // The positions idx0, idx0 + stride, idx0 + 2 stride, ... are visited IN ORDER, and the stride is a whole number of pixels
  // (a multiple of c8): (n, y, x) of a position follow from the previous one by constant increments with two carries --
  // the three 32-bit divisions per 16-byte element of the first form were more instructions than the rest of the pass
  // (bn_reduce<0> on a 0.6 GB map: 165 us, ALU-bound).
  int wk_x, wk_y, wk_n;
  {
    const int p0 = ((int)blockIdx.x * 256 + (int)threadIdx.x) / c8, row0 = p0 / W;
    wk_x = p0 - row0 * W, wk_n = row0 / H, wk_y = row0 - wk_n * H;
  }
  const int wk_dp = stride / c8, wk_drow = wk_dp / W, wk_dx = wk_dp - wk_drow * W, wk_dn = wk_drow / H, wk_dy = wk_drow - wk_dn * H;
  auto next_offset = [&]() -> int64_t {
    const int64_t off = (((int64_t)wk_n * (H + 2) + wk_y + 1) * (W + 2) + wk_x + 1) * C + cg * 8;
    wk_x += wk_dx;
    const int cx = wk_x >= W ? 1 : 0;
    wk_x -= cx ? W : 0;
    wk_y += wk_dy + cx;
    const int cy = wk_y >= H ? 1 : 0;
    wk_y -= cy ? H : 0;
    wk_n += wk_dn + cy;
    return off;
  };
  const u32x4 ones = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  auto apply = [&](const u32x4 zv, const u32x4 gv, const u32x4 yv) {
    u32x4 o;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      float v[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int e = 2 * w + h;
        const float zz = h ? bf_hi(zv[w]) : bf_lo(zv[w]);
        const float yy_ = MASKZ ? fmaf(zz, sc[e], sh[e]) : (h ? bf_hi(yv[w]) : bf_lo(yv[w]));
        const float g = yy_ > 0.f ? (h ? bf_hi(gv[w]) : bf_lo(gv[w])) : 0.f;
        const float zh = (zz - mu[e]) * is[e];
        v[h] = gk[e] * is[e] * (g - db[e] - zh * dg[e]);
      }
      o[w] = pack_bf16x2(v[0], v[1]);
    }
    return o;
  };
  int idx = (int)blockIdx.x * 256 + (int)threadIdx.x;
  for (; idx + stride < total; idx += 2 * stride) {
    int64_t off[2];
    u32x4 zv[2], gv[2], yv[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      off[q] = next_offset();
      zv[q] = *reinterpret_cast<const u32x4*>(z + off[q]);
      gv[q] = *reinterpret_cast<const u32x4*>(dy + off[q]);
      yv[q] = (y && !MASKZ) ? *reinterpret_cast<const u32x4*>(y + off[q]) : ones;
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) *reinterpret_cast<u32x4*>(dz + off[q]) = apply(zv[q], gv[q], yv[q]);
  }
  for (; idx < total; idx += stride) {
    const int64_t off = next_offset();
    *reinterpret_cast<u32x4*>(dz + off) = apply(*reinterpret_cast<const u32x4*>(z + off), *reinterpret_cast<const u32x4*>(dy + off),
                                                (y && !MASKZ) ? *reinterpret_cast<const u32x4*>(y + off) : ones);
  }
}

// The two APPLY passes of the batch norm behind the x8 deconv, eight output pixels per thread: the pixels 8 k - 4 .. 8 k + 3 of
// an output row share their four source vectors (one phase group of the deconv), so a thread loads and unpacks them once and
// produces eight 16-byte results with compile-time column weights -- a per-element form (upsample_words per result inside the
// grid-stride kernels above) spent ~150 instructions per result on its loads, address arithmetic and weights and was ALU-bound:
// no faster than reading the stored map.  Same expressions per value (upsample_words' fmaf chain, the apply lambdas of the kernels above): same bits.
// Grid: x over (phase group k = 0 .. W/8, channel group), y strides over the N H output rows (a thread keeps its channels'
// constants for all its rows).
template <bool BWD>
__global__ __launch_bounds__(256) void bn_ups8_apply8_kernel(const __bf16* __restrict__ low, const __bf16* __restrict__ dy,
                                                            const float* __restrict__ mean, const float* __restrict__ invstd,
                                                            const float* __restrict__ gamma, const double* __restrict__ sums,
                                                            double M, const float* __restrict__ zsc, const float* __restrict__ zsh,
                                                            int relu, __bf16* __restrict__ out, int N, int H, int W, int C) {
  const int c8 = C >> 3, Hi = H >> 3, Wi = W >> 3;
  const int idx = (int)blockIdx.x * 256 + (int)threadIdx.x;
  if (idx >= (Wi + 1) * c8) return;
  const int k = idx / c8, cg = idx - k * c8;
  float mu[8], is[8], gk[8], db[8], dg[8], sc[8], sh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = cg * 8 + e;
    sc[e] = zsc[c], sh[e] = zsh[c];
    mu[e] = BWD ? mean[c] : 0.f, is[e] = BWD ? invstd[c] : 0.f, gk[e] = BWD ? gamma[c] : 0.f;
    db[e] = BWD ? (float)(sums[c] / M) : 0.f, dg[e] = BWD ? (float)(sums[C + c] / M) : 0.f;
  }
  const int64_t rowp = (int64_t)(Wi + 2) * C;
  for (int row = blockIdx.y; row < N * H; row += gridDim.y) {
    const int n = row / H, yy = row - n * H;
    const int iy1 = (yy + 4) >> 3, py = (yy + 4) & 7;
    const float wy1 = (float)(2 * py + 1) * 0.0625f, wy0 = (float)(15 - 2 * py) * 0.0625f;
    const __bf16* p00 = low + (((int64_t)n * (Hi + 2) + iy1) * (Wi + 2) + k) * C + cg * 8;
    const u32x4 a00 = *reinterpret_cast<const u32x4*>(p00), a01 = *reinterpret_cast<const u32x4*>(p00 + C);
    const u32x4 a10 = *reinterpret_cast<const u32x4*>(p00 + rowp), a11 = *reinterpret_cast<const u32x4*>(p00 + rowp + C);
    const int64_t orow = ((int64_t)n * (H + 2) + yy + 1) * (W + 2) + 1;
    u32x4 gv[8];
    if constexpr (BWD) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int ox = 8 * k - 4 + j;
        if (ox >= 0 && ox < W) gv[j] = *reinterpret_cast<const u32x4*>(dy + (orow + ox) * C + cg * 8);
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int ox = 8 * k - 4 + j;
      if (ox < 0 || ox >= W) continue;
      const float wx1 = (float)(2 * j + 1) * 0.0625f, wx0 = (float)(15 - 2 * j) * 0.0625f;  // ((ox + 4) & 7 == j)
      const float w00 = wy0 * wx0, w01 = wy0 * wx1, w10 = wy1 * wx0, w11 = wy1 * wx1;
      u32x4 zv;
#pragma unroll
      for (int w = 0; w < 4; ++w)
        zv[w] = pack_bf16x2(fmaf(bf_lo(a11[w]), w11, fmaf(bf_lo(a10[w]), w10, fmaf(bf_lo(a01[w]), w01, bf_lo(a00[w]) * w00))),
                            fmaf(bf_hi(a11[w]), w11, fmaf(bf_hi(a10[w]), w10, fmaf(bf_hi(a01[w]), w01, bf_hi(a00[w]) * w00))));
      u32x4 o;
      if constexpr (!BWD) {
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          float a = fmaf(bf_lo(zv[w]), sc[2 * w], sh[2 * w]);
          float b = fmaf(bf_hi(zv[w]), sc[2 * w + 1], sh[2 * w + 1]);
          if (relu) {
            a = a > 0.f ? a : 0.f;
            b = b > 0.f ? b : 0.f;
          }
          o[w] = pack_bf16x2(a, b);
        }
      } else {
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          float v[2];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int e = 2 * w + h;
            const float zz = h ? bf_hi(zv[w]) : bf_lo(zv[w]);
            const float yy_ = fmaf(zz, sc[e], sh[e]);
            const float g = yy_ > 0.f ? (h ? bf_hi(gv[j][w]) : bf_lo(gv[j][w])) : 0.f;
            const float zh = (zz - mu[e]) * is[e];
            v[h] = gk[e] * is[e] * (g - db[e] - zh * dg[e]);
          }
          o[w] = pack_bf16x2(v[0], v[1]);
        }
      }
      *reinterpret_cast<u32x4*>(out + (orow + ox) * C + cg * 8) = o;
    }
  }
}

// ... and the two REDUCE passes in the same thread layout (BWD = false: sum z, sum z^2; true: sum g, sum g zhat with the relu
// mask from z -- bn_reduce_kernel's MODE 0 / 2).  fp32 per thread over its rows, the workgroup through LDS, one row of `part`
// per workgroup (blockIdx.y * gridDim.x + blockIdx.x) for bn_sums_kernel / bn_sums_finalize_kernel: a fixed order, bitwise
// reproducible -- but a DIFFERENT order of the same terms than bn_reduce_kernel's, so the sums agree with the stored-map
// path to fp32 rounding, not bit for bit.
template <bool BWD>
__global__ __launch_bounds__(256) void bn_ups8_reduce8_kernel(const __bf16* __restrict__ low, const __bf16* __restrict__ dy,
                                                             const float* __restrict__ mean, const float* __restrict__ invstd,
                                                             const float* __restrict__ zsc, const float* __restrict__ zsh,
                                                             float* __restrict__ part, int N, int H, int W, int C) {
  const int c8 = C >> 3, Hi = H >> 3, Wi = W >> 3;
  const int idx = (int)blockIdx.x * 256 + (int)threadIdx.x;
  const bool live = idx < (Wi + 1) * c8;
  const int k = live ? idx / c8 : 0, cg = threadIdx.x % c8;  // (256 and blockIdx.x * 256 are multiples of c8)
  float s0[8], s1[8], mu[8], is[8], sc[8], sh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = cg * 8 + e;
    s0[e] = s1[e] = 0.f;
    mu[e] = BWD ? mean[c] : 0.f, is[e] = BWD ? invstd[c] : 0.f, sc[e] = BWD ? zsc[c] : 0.f, sh[e] = BWD ? zsh[c] : 0.f;
  }
  const int64_t rowp = (int64_t)(Wi + 2) * C;
  for (int row = blockIdx.y; live && row < N * H; row += gridDim.y) {
    const int n = row / H, yy = row - n * H;
    const int iy1 = (yy + 4) >> 3, py = (yy + 4) & 7;
    const float wy1 = (float)(2 * py + 1) * 0.0625f, wy0 = (float)(15 - 2 * py) * 0.0625f;
    const __bf16* p00 = low + (((int64_t)n * (Hi + 2) + iy1) * (Wi + 2) + k) * C + cg * 8;
    const u32x4 a00 = *reinterpret_cast<const u32x4*>(p00), a01 = *reinterpret_cast<const u32x4*>(p00 + C);
    const u32x4 a10 = *reinterpret_cast<const u32x4*>(p00 + rowp), a11 = *reinterpret_cast<const u32x4*>(p00 + rowp + C);
    const int64_t orow = ((int64_t)n * (H + 2) + yy + 1) * (W + 2) + 1;
    u32x4 gv[8];
    if constexpr (BWD) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int ox = 8 * k - 4 + j;
        if (ox >= 0 && ox < W) gv[j] = *reinterpret_cast<const u32x4*>(dy + (orow + ox) * C + cg * 8);
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int ox = 8 * k - 4 + j;
      if (ox < 0 || ox >= W) continue;
      const float wx1 = (float)(2 * j + 1) * 0.0625f, wx0 = (float)(15 - 2 * j) * 0.0625f;
      const float w00 = wy0 * wx0, w01 = wy0 * wx1, w10 = wy1 * wx0, w11 = wy1 * wx1;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const uint32_t zw = pack_bf16x2(fmaf(bf_lo(a11[w]), w11, fmaf(bf_lo(a10[w]), w10, fmaf(bf_lo(a01[w]), w01, bf_lo(a00[w]) * w00))),
                                        fmaf(bf_hi(a11[w]), w11, fmaf(bf_hi(a10[w]), w10, fmaf(bf_hi(a01[w]), w01, bf_hi(a00[w]) * w00))));
        const float za = bf_lo(zw), zb = bf_hi(zw);
        if constexpr (!BWD) {
          s0[2 * w] += za;
          s1[2 * w] += za * za;
          s0[2 * w + 1] += zb;
          s1[2 * w + 1] += zb * zb;
        } else {
          const float ga = fmaf(za, sc[2 * w], sh[2 * w]) > 0.f ? bf_lo(gv[j][w]) : 0.f;
          const float gb = fmaf(zb, sc[2 * w + 1], sh[2 * w + 1]) > 0.f ? bf_hi(gv[j][w]) : 0.f;
          s0[2 * w] += ga;
          s1[2 * w] += ga * (za - mu[2 * w]) * is[2 * w];
          s0[2 * w + 1] += gb;
          s1[2 * w + 1] += gb * (zb - mu[2 * w + 1]) * is[2 * w + 1];
        }
      }
    }
  }
  __shared__ float red[256][17];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    red[threadIdx.x][e] = s0[e];
    red[threadIdx.x][8 + e] = s1[e];
  }
  __syncthreads();
  const int64_t prow = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
  for (int t = threadIdx.x; t < c8 * 16; t += 256) {
    const int g = t >> 4, col = t & 15;
    float acc = 0.f;
    for (int r = g; r < 256; r += c8) acc += red[r][col];
    const int ch = g * 8 + (col & 7);
    part[prow * 2 * C + (col < 8 ? ch : C + ch)] = acc;
  }
}

// y = relu(z * scale + shift) AND its 2x2 max-pool in one pass over the POOLED pixels (a thread owns one 8-channel group):
// bn_apply + maxpool_kernel read the activation map back once more; y == nullptr skips the full-resolution map
// altogether (training: the gradient passes recompute it from z, bn_pool_bwd_kernel)
__global__ __launch_bounds__(256) void bn_apply_pool_kernel(const __bf16* __restrict__ z, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, __bf16* __restrict__ y,
                                                           __bf16* __restrict__ pooled, int N, int Ho, int Wo, int C) {
  const int c8 = C >> 3;
  const int cg = threadIdx.x % c8;
  float sc[8], sh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) sc[e] = scale[cg * 8 + e], sh[e] = shift[cg * 8 + e];
  const int Hi = 2 * Ho, Wi = 2 * Wo;
  const int64_t rowp = (int64_t)(Wi + 2) * C;
  const int total = N * Ho * Wo * c8, stride = (int)gridDim.x * 256;
  for (int idx = (int)blockIdx.x * 256 + (int)threadIdx.x; idx < total; idx += stride) {
    const int p = idx / c8;
    const int row = p / Wo;
    const int ox = p - row * Wo, n = row / Ho, oy = row - n * Ho;
    const int64_t off = (((int64_t)n * (Hi + 2) + (2 * oy + 1)) * (Wi + 2) + (2 * ox + 1)) * C + cg * 8;
    const u32x4 v[4] = {*reinterpret_cast<const u32x4*>(z + off), *reinterpret_cast<const u32x4*>(z + off + C),
                        *reinterpret_cast<const u32x4*>(z + off + rowp), *reinterpret_cast<const u32x4*>(z + off + rowp + C)};
    u32x4 o[4], q;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      float m0 = 0.f, m1 = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float a = fmaf(bf_lo(v[k][w]), sc[2 * w], sh[2 * w]);
        float b = fmaf(bf_hi(v[k][w]), sc[2 * w + 1], sh[2 * w + 1]);
        a = a > 0.f ? a : 0.f;
        b = b > 0.f ? b : 0.f;
        o[k][w] = pack_bf16x2(a, b);
        m0 = fmaxf(m0, bf_lo(o[k][w]));  // the max of the ROUNDED activations (what maxpool_kernel sees); all >= 0
        m1 = fmaxf(m1, bf_hi(o[k][w]));
      }
      q[w] = pack_bf16x2(m0, m1);
    }
    if (y != nullptr) {
      *reinterpret_cast<u32x4*>(y + off) = o[0];
      *reinterpret_cast<u32x4*>(y + off + C) = o[1];
      *reinterpret_cast<u32x4*>(y + off + rowp) = o[2];
      *reinterpret_cast<u32x4*>(y + off + rowp + C) = o[3];
    }
    *reinterpret_cast<u32x4*>(pooled + (((int64_t)n * (Ho + 2) + (oy + 1)) * (Wo + 2) + (ox + 1)) * C + cg * 8) = q;
  }
}

// ---- MaxPoolGrad + ReluGrad + the batch-norm gradient of the conv in front of a 2x2 max-pool, in the two normalisation
// passes themselves (xv_maxpool2x2_bwd wrote the routed gradient map, both passes then read it back: 3 of 7 streams).
// A thread owns one 8-channel group and walks the POOLED pixels: it recomputes its window's four activations
// y = relu(bf16(z * scale + shift)) -- the forward pass's own expression and rounding --, routes the pooled gradient to
// the first maximum if that is positive (maxpool_bwd_kernel's rule, on the same values), and either accumulates
// sum g / sum g * zhat (APPLY = 0) or writes dz = gamma * invstd * (g - dbeta / M - zhat * dgamma / M) for all four
// positions (APPLY = 1).
template <int APPLY>
__global__ __launch_bounds__(256) void bn_pool_bwd_kernel(const __bf16* __restrict__ dp, const __bf16* __restrict__ z,
                                                         const float* __restrict__ mean, const float* __restrict__ invstd,
                                                         const float* __restrict__ scale, const float* __restrict__ shift,
                                                         const float* __restrict__ gamma, double* __restrict__ sums, double M,
                                                         __bf16* __restrict__ dz, int N, int Ho, int Wo, int C,
                                                         float* __restrict__ part) {
  const int c8 = C >> 3;
  const int cg = threadIdx.x % c8;  // constant over the loop: 256 and the grid stride are multiples of c8
  float mu[8], is[8], sc[8], sh[8], gk[8], db[8], dg[8], s0[8], s1[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = cg * 8 + e;
    mu[e] = mean[c], is[e] = invstd[c], sc[e] = scale[c], sh[e] = shift[c];
    s0[e] = s1[e] = 0.f;
    gk[e] = APPLY ? gamma[c] : 0.f;
    db[e] = APPLY ? (float)(sums[c] / M) : 0.f;
    dg[e] = APPLY ? (float)(sums[C + c] / M) : 0.f;
  }
  const int Hi = 2 * Ho, Wi = 2 * Wo;
  const int64_t rowp = (int64_t)(Wi + 2) * C;
  const int total = N * Ho * Wo * c8, stride = (int)gridDim.x * 256;  // < 2^31 (checked by the launcher)
  for (int idx = (int)blockIdx.x * 256 + (int)threadIdx.x; idx < total; idx += stride) {
    const int p = idx / c8;
    const int row = p / Wo;
    const int ox = p - row * Wo, n = row / Ho, oy = row - n * Ho;
    const int64_t off = (((int64_t)n * (Hi + 2) + (2 * oy + 1)) * (Wi + 2) + (2 * ox + 1)) * C + cg * 8;
    const u32x4 g = *reinterpret_cast<const u32x4*>(dp + (((int64_t)n * (Ho + 2) + (oy + 1)) * (Wo + 2) + (ox + 1)) * C + cg * 8);
    const u32x4 v[4] = {*reinterpret_cast<const u32x4*>(z + off), *reinterpret_cast<const u32x4*>(z + off + C),
                        *reinterpret_cast<const u32x4*>(z + off + rowp), *reinterpret_cast<const u32x4*>(z + off + rowp + C)};
    u32x4 o[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      float out[4][2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int e = 2 * w + h;
        float zz[4], yy[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          zz[k] = h ? bf_hi(v[k][w]) : bf_lo(v[k][w]);
          const float a = fmaf(zz[k], sc[e], sh[e]);
          yy[k] = bf_lo(pack_bf16x2(a > 0.f ? a : 0.f, 0.f));  // the stored activation: relu, rounded to bf16
        }
        int best = 0;
        float m = yy[0];
#pragma unroll
        for (int k = 1; k < 4; ++k)
          if (yy[k] > m) {
            m = yy[k];
            best = k;
          }
        const float gp = m > 0.f ? (h ? bf_hi(g[w]) : bf_lo(g[w])) : 0.f;
        if (!APPLY) {
          float zb = zz[0];
#pragma unroll
          for (int k = 1; k < 4; ++k) zb = best == k ? zz[k] : zb;
          s0[e] += gp;
          s1[e] += gp * (zb - mu[e]) * is[e];
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float zh = (zz[k] - mu[e]) * is[e];
            out[k][h] = gk[e] * is[e] * ((k == best ? gp : 0.f) - db[e] - zh * dg[e]);
          }
        }
      }
      if (APPLY) {
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k][w] = pack_bf16x2(out[k][0], out[k][1]);
      }
    }
    if (APPLY) {
      *reinterpret_cast<u32x4*>(dz + off) = o[0];
      *reinterpret_cast<u32x4*>(dz + off + C) = o[1];
      *reinterpret_cast<u32x4*>(dz + off + rowp) = o[2];
      *reinterpret_cast<u32x4*>(dz + off + rowp + C) = o[3];
    }
  }
  if (APPLY) return;
  __shared__ float red[256][17];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    red[threadIdx.x][e] = s0[e];
    red[threadIdx.x][8 + e] = s1[e];
  }
  __syncthreads();
  for (int t = threadIdx.x; t < c8 * 16; t += 256) {
    const int gq = t >> 4, col = t & 15;
    float acc = 0.f;
    for (int row = gq; row < 256; row += c8) acc += red[row][col];
    const int ch = gq * 8 + (col & 7);
    if (part != nullptr)
      part[(int64_t)blockIdx.x * 2 * C + (col < 8 ? ch : C + ch)] = acc;
    else
      atomicAdd(sums + (col < 8 ? ch : C + ch), (double)acc);
  }
}

// dgamma[c] += sums[C + c], dbeta[c] += sums[c]
__global__ void bn_grads_kernel(const double* __restrict__ sums, int C, float* __restrict__ dgamma,
                                float* __restrict__ dbeta) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  dbeta[c] += (float)sums[c];
  dgamma[c] += (float)sums[C + c];
}

// ---- dense float32 [M][C] forms (the batch norm on `score`, C <= 32) ------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void bn_dense_reduce_kernel(const float* __restrict__ z, const float* __restrict__ dy,
                                                             const float* __restrict__ mean,
                                                             const float* __restrict__ invstd,
                                                             double* __restrict__ sums, int64_t M, int C,
                                                             float* __restrict__ part = nullptr) {
  __shared__ float red[4][64];
  float s0[32], s1[32];
#pragma unroll
  for (int c = 0; c < 32; ++c) s0[c] = s1[c] = 0.f;
  auto accumulate = [&](int c, float v, float g) {
    if (MODE == 0) {
      s0[c] += v;
      s1[c] += v * v;
    } else {
      s0[c] += g;
      s1[c] += g * (v - mean[c]) * invstd[c];
    }
  };
  if ((C & 3) == 0) {
    // rows of C floats are 16-byte multiples: a lane reads its row as C/4 vector loads, the wave a dense span (with
    // scalar loads every instruction touched 24 cache lines for 256 bytes and the kernel ran at 1 TB/s)
    const int64_t stride = (int64_t)gridDim.x * 256;
    int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (C <= 16) {
      // four rows' loads in flight per thread, accumulated in row order (the sums do not depend on the unrolling): with one
      // row -- three 16-byte loads -- in flight the pass ran at 2.7 TB/s
      for (; p + 3 * stride < M; p += 4 * stride) {
        f32x4 v[4][4], g[4][4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int c4 = 0; c4 < 4; ++c4)
            if (c4 * 4 < C) {
              v[q][c4] = *reinterpret_cast<const f32x4*>(z + (p + q * stride) * C + c4 * 4);
              if (MODE == 1) g[q][c4] = *reinterpret_cast<const f32x4*>(dy + (p + q * stride) * C + c4 * 4);
            }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int c4 = 0; c4 < 4; ++c4)
            if (c4 * 4 < C) {
              const f32x4 gg = MODE == 1 ? g[q][c4] : f32x4{0.f, 0.f, 0.f, 0.f};
              accumulate(c4 * 4, v[q][c4].x, gg.x);
              accumulate(c4 * 4 + 1, v[q][c4].y, gg.y);
              accumulate(c4 * 4 + 2, v[q][c4].z, gg.z);
              accumulate(c4 * 4 + 3, v[q][c4].w, gg.w);
            }
      }
    }
    for (; p < M; p += stride) {
#pragma unroll
      for (int c4 = 0; c4 < 8; ++c4)
        if (c4 * 4 < C) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(z + p * C + c4 * 4);
          f32x4 g = f32x4{0.f, 0.f, 0.f, 0.f};
          if (MODE == 1) g = *reinterpret_cast<const f32x4*>(dy + p * C + c4 * 4);
          accumulate(c4 * 4, v.x, g.x);
          accumulate(c4 * 4 + 1, v.y, g.y);
          accumulate(c4 * 4 + 2, v.z, g.z);
          accumulate(c4 * 4 + 3, v.w, g.w);
        }
    }
  } else {
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < M; p += (int64_t)gridDim.x * 256) {
#pragma unroll
      for (int c = 0; c < 32; ++c)
        if (c < C) accumulate(c, z[p * C + c], MODE == 1 ? dy[p * C + c] : 0.f);
    }
  }
  // a butterfly inside each wave, the four waves in wave order: a fixed tree (LDS float atomics added in arrival order)
#pragma unroll
  for (int c = 0; c < 32; ++c)
    if (c < C) {
      float a = s0[c], b = s1[c];
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) {
        a += __shfl_xor(a, off, 64);
        b += __shfl_xor(b, off, 64);
      }
      if ((threadIdx.x & 63) == 0) {
        red[threadIdx.x >> 6][c] = a;
        red[threadIdx.x >> 6][32 + c] = b;
      }
    }
  __syncthreads();
  if ((int)threadIdx.x < C) {
    const int c = threadIdx.x;
    const float a = ((red[0][c] + red[1][c]) + red[2][c]) + red[3][c];
    const float b = ((red[0][32 + c] + red[1][32 + c]) + red[2][32 + c]) + red[3][32 + c];
    if (part != nullptr) {
      part[(int64_t)blockIdx.x * 2 * C + c] = a;
      part[(int64_t)blockIdx.x * 2 * C + C + c] = b;
    } else {
      atomicAdd(sums + c, (double)a);
      atomicAdd(sums + C + c, (double)b);
    }
  }
}

__global__ __launch_bounds__(256) void bn_dense_apply_kernel(const float* __restrict__ z, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, float* __restrict__ y,
                                                            int64_t total, int C) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    y[i] = z[i] * scale[c] + shift[c];
  }
}

// The same pass for C == CM, a multiple of four: a thread owns whole rows (16-byte loads / stores), the per-channel constants
// -- computed once per workgroup, through LDS -- in 5 CM registers.  (The per-float form below pays a 64-bit modulo, two
// double divisions and three constant loads per float: 172 us for 0.7 GB; a first row form with constants for up to 32
// classes per thread lost more to its 160 registers than it gained.)
template <int CM>
__global__ __launch_bounds__(256) void bn_dense_bwd_apply_rows_kernel(const float* __restrict__ dy, const float* __restrict__ z,
                                                                     const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                     const float* __restrict__ gamma, const double* __restrict__ sums,
                                                                     double M, float* __restrict__ dz, int64_t rows) {
  __shared__ float cst[5][CM];
  if ((int)threadIdx.x < CM) {
    const int c = threadIdx.x;
    cst[0][c] = mean[c], cst[1][c] = invstd[c], cst[2][c] = gamma[c] * invstd[c];
    cst[3][c] = (float)(sums[c] / M), cst[4][c] = (float)(sums[CM + c] / M);
  }
  __syncthreads();
  float mu[CM], is[CM], gk[CM], db[CM], dg[CM];
#pragma unroll
  for (int c = 0; c < CM; ++c) mu[c] = cst[0][c], is[c] = cst[1][c], gk[c] = cst[2][c], db[c] = cst[3][c], dg[c] = cst[4][c];
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < rows; p += (int64_t)gridDim.x * 256) {
    f32x4 g[CM / 4], v[CM / 4];
#pragma unroll
    for (int c4 = 0; c4 < CM / 4; ++c4) {
      g[c4] = *reinterpret_cast<const f32x4*>(dy + p * CM + c4 * 4);
      v[c4] = *reinterpret_cast<const f32x4*>(z + p * CM + c4 * 4);
    }
#pragma unroll
    for (int c4 = 0; c4 < CM / 4; ++c4) {
      f32x4 o;
      o.x = gk[c4 * 4] * (g[c4].x - db[c4 * 4] - (v[c4].x - mu[c4 * 4]) * is[c4 * 4] * dg[c4 * 4]);
      o.y = gk[c4 * 4 + 1] * (g[c4].y - db[c4 * 4 + 1] - (v[c4].y - mu[c4 * 4 + 1]) * is[c4 * 4 + 1] * dg[c4 * 4 + 1]);
      o.z = gk[c4 * 4 + 2] * (g[c4].z - db[c4 * 4 + 2] - (v[c4].z - mu[c4 * 4 + 2]) * is[c4 * 4 + 2] * dg[c4 * 4 + 2]);
      o.w = gk[c4 * 4 + 3] * (g[c4].w - db[c4 * 4 + 3] - (v[c4].w - mu[c4 * 4 + 3]) * is[c4 * 4 + 3] * dg[c4 * 4 + 3]);
      *reinterpret_cast<f32x4*>(dz + p * CM + c4 * 4) = o;
    }
  }
}

__global__ __launch_bounds__(256) void bn_dense_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ z,
                                                                const float* __restrict__ mean,
                                                                const float* __restrict__ invstd,
                                                                const float* __restrict__ gamma,
                                                                const double* __restrict__ sums, double M,
                                                                float* __restrict__ dz, int64_t total, int C) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    const float zh = (z[i] - mean[c]) * invstd[c];
    dz[i] = gamma[c] * invstd[c] * (dy[i] - (float)(sums[c] / M) - zh * (float)(sums[C + c] / M));
  }
}

// ---- raw bilinear up-sampling (the constant deconv kernel, custom_layers.py:8-25) and its transpose -----------

template <int S>
__global__ __launch_bounds__(256) void upsample_raw_kernel(const __bf16* __restrict__ x, __bf16* __restrict__ y, int N,
                                                          int Hi, int Wi, int C) {
  const int c8 = C >> 3;
  const int Ho = Hi * S, Wo = Wi * S;
  // (32-bit index arithmetic: the launcher checks total < 2^31; the 64-bit divisions of the first form cost more than the
  // kernel's memory traffic -- 215 us for a 0.6 GB store)
  // grid: x over (output column, channel group) of one output row, y = output row, z = image -- no division per element
  // (the flat index form spent more on 64-bit divisions than on its memory traffic: 215 us for a 0.6 GB store)
  const int oy = blockIdx.y, n = blockIdx.z;
  for (uint32_t idx = blockIdx.x * 256 + threadIdx.x; idx < (uint32_t)(Wo * c8); idx += gridDim.x * 256) {
    const int ox = (int)(idx / (uint32_t)c8), cg = (int)(idx - (uint32_t)ox * c8);
    const u32x4 o = upsample_words<S>(x, n, oy, ox, cg, Hi, Wi, C);
    *reinterpret_cast<u32x4*>(y + (((int64_t)n * (Ho + 2) + oy + 1) * (Wo + 2) + ox + 1) * C + cg * 8) = o;
  }
}

// dx[i, j] = sum over the 2S x 2S output footprint of w(oy, i) * w(ox, j) * dy[oy, ox]
template <int S>
__global__ __launch_bounds__(256) void upsample_raw_bwd_kernel(const __bf16* __restrict__ dy, __bf16* __restrict__ dx,
                                                              int N, int Hi, int Wi, int C) {
  const int c8 = C >> 3;
  const int Ho = Hi * S, Wo = Wi * S;
  const uint32_t total = (uint32_t)N * Hi * Wi * c8;  // (< 2^31: checked by the launcher)
  for (uint32_t idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
    const int cg = (int)(idx % (uint32_t)c8);
    uint32_t r = idx / (uint32_t)c8;
    const int j = (int)(r % (uint32_t)Wi);
    r /= (uint32_t)Wi;
    const int i = (int)(r % (uint32_t)Hi);
    const int n = (int)(r / (uint32_t)Hi);
    const __bf16* dimg = dy + (int64_t)n * (Ho + 2) * (Wo + 2) * C + cg * 8;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // A row of the footprint = 2S loads issued together (positions outside the map read the zero border of the padded
    // buffer or a clamped address with weight 0): with one load in flight per thread the 256-tap gather of the x8 form was
    // a chain of 256 memory latencies (422 us for 0.6 GB)
    for (int oy = i * S - S / 2; oy < i * S - S / 2 + 2 * S; ++oy) {
      if (oy < 0 || oy >= Ho) continue;
      const float wy = bl_w<S>(oy, i);
      const __bf16* rowp = dimg + (int64_t)(oy + 1) * (Wo + 2) * C;
      u32x4 gv[2 * S];
      float wx[2 * S];
#pragma unroll
      for (int t = 0; t < 2 * S; ++t) {
        const int ox = j * S - S / 2 + t;
        const bool in = ox >= 0 && ox < Wo;
        wx[t] = in ? wy * bl_w<S>(ox, j) : 0.f;
        gv[t] = *reinterpret_cast<const u32x4*>(rowp + (int64_t)((in ? ox : 0) + 1) * C);
      }
#pragma unroll
      for (int t = 0; t < 2 * S; ++t)
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          acc[2 * w] += wx[t] * bf_lo(gv[t][w]);
          acc[2 * w + 1] += wx[t] * bf_hi(gv[t][w]);
        }
    }
    u32x4 o;
#pragma unroll
    for (int w = 0; w < 4; ++w) o[w] = pack_bf16x2(acc[2 * w], acc[2 * w + 1]);
    *reinterpret_cast<u32x4*>(dx + (((int64_t)n * (Hi + 2) + i + 1) * (Wi + 2) + j + 1) * C + cg * 8) = o;
  }
}

// The x8 form without the gather's read amplification (upsample_raw_bwd_kernel<8> reads every gradient element from four
// source pixels' footprints: 256 16-byte loads per output, 262 us for a 0.6 GB map).  The outputs 8 k - 4 .. 8 k + 3 of a row /
// column form a phase group whose two sources are k - 1 (weights w0(p) = (15 - 2 p) / 16) and k (w1(p) = (2 p + 1) / 16), so
//   dx[i][j] = S11(B(i, j)) + S10(B(i, j + 1)) + S01(B(i + 1, j)) + S00(B(i + 1, j + 1)),
//   S_ab(B) = sum over the 8x8 block B(ky, kx) of w_a(p) w_b(q) dy[8 ky - 4 + p][8 kx - 4 + q].
// Pass 1: one thread per block and channel group reads its 64 elements ONCE and leaves the four sums (fp32) in the workspace
// S[4][N][Hi + 1][Wi + 1][C]; pass 2 adds the four terms of every source pixel in that fixed order and rounds to bf16.
__global__ __launch_bounds__(256) void upsample8_bwd_blocks_kernel(const __bf16* __restrict__ dy, float* __restrict__ S, int N, int Hi,
                                                                  int Wi, int C) {
  const int c8 = C >> 3, Ho = 8 * Hi, Wo = 8 * Wi;
  const int idx = (int)blockIdx.x * 256 + (int)threadIdx.x;
  if (idx >= (Wi + 1) * c8) return;
  const int kx = idx / c8, cg = idx - kx * c8, ky = blockIdx.y, n = blockIdx.z;
  float s11[8], s10[8], s01[8], s00[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s11[e] = s10[e] = s01[e] = s00[e] = 0.f;
  const __bf16* dimg = dy + (int64_t)n * (Ho + 2) * (Wo + 2) * C + cg * 8;
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int oy = 8 * ky - 4 + p;
    if (oy < 0 || oy >= Ho) continue;
    const __bf16* rowp = dimg + ((int64_t)(oy + 1) * (Wo + 2) + 1) * C;
    u32x4 gv[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int ox = 8 * kx - 4 + q;
      gv[q] = (ox >= 0 && ox < Wo) ? *reinterpret_cast<const u32x4*>(rowp + (int64_t)ox * C) : u32x4{0u, 0u, 0u, 0u};
    }
    float c1[8], c0[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) c1[e] = c0[e] = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float w1 = (float)(2 * q + 1) * 0.0625f, w0 = (float)(15 - 2 * q) * 0.0625f;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const float a = bf_lo(gv[q][w]), b = bf_hi(gv[q][w]);
        c1[2 * w] = fmaf(a, w1, c1[2 * w]), c0[2 * w] = fmaf(a, w0, c0[2 * w]);
        c1[2 * w + 1] = fmaf(b, w1, c1[2 * w + 1]), c0[2 * w + 1] = fmaf(b, w0, c0[2 * w + 1]);
      }
    }
    const float v1 = (float)(2 * p + 1) * 0.0625f, v0 = (float)(15 - 2 * p) * 0.0625f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      s11[e] = fmaf(c1[e], v1, s11[e]), s10[e] = fmaf(c0[e], v1, s10[e]);
      s01[e] = fmaf(c1[e], v0, s01[e]), s00[e] = fmaf(c0[e], v0, s00[e]);
    }
  }
  const int64_t plane = (int64_t)N * (Hi + 1) * (Wi + 1) * C;
  float* dst = S + (((int64_t)n * (Hi + 1) + ky) * (Wi + 1) + kx) * C + cg * 8;
  auto put = [&](float* d, const float (&v)[8]) {
    *reinterpret_cast<f32x4*>(d) = f32x4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(d + 4) = f32x4{v[4], v[5], v[6], v[7]};
  };
  put(dst, s11), put(dst + plane, s10), put(dst + 2 * plane, s01), put(dst + 3 * plane, s00);
}

__global__ __launch_bounds__(256) void upsample8_bwd_combine_kernel(const float* __restrict__ S, __bf16* __restrict__ dx, int N, int Hi,
                                                                   int Wi, int C) {
  const int c8 = C >> 3;
  const int idx = (int)blockIdx.x * 256 + (int)threadIdx.x;
  if (idx >= Wi * c8) return;
  const int j = idx / c8, cg = idx - j * c8, i = blockIdx.y, n = blockIdx.z;
  const int64_t plane = (int64_t)N * (Hi + 1) * (Wi + 1) * C;
  const float* b00 = S + (((int64_t)n * (Hi + 1) + i) * (Wi + 1) + j) * C + cg * 8;
  const int64_t dxn = C, dyn = (int64_t)(Wi + 1) * C;
  u32x4 o;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(b00 + 4 * h);                          // S11 of block (i, j)
    const f32x4 b = *reinterpret_cast<const f32x4*>(b00 + plane + dxn + 4 * h);            // S10 of block (i, j + 1)
    const f32x4 c = *reinterpret_cast<const f32x4*>(b00 + 2 * plane + dyn + 4 * h);        // S01 of block (i + 1, j)
    const f32x4 d = *reinterpret_cast<const f32x4*>(b00 + 3 * plane + dyn + dxn + 4 * h);  // S00 of block (i + 1, j + 1)
    const f32x4 t = ((a + b) + c) + d;
    o[2 * h] = pack_bf16x2(t.x, t.y);
    o[2 * h + 1] = pack_bf16x2(t.z, t.w);
  }
  *reinterpret_cast<u32x4*>(dx + (((int64_t)n * (Hi + 2) + i + 1) * (Wi + 2) + j + 1) * C + cg * 8) = o;
}

// ---- per-pixel U -> C score conv on the full-resolution map, dense float32 output -------------------------------
// A thread owns a pixel (its C sums stay in registers).  The data gradient below goes through an LDS tile so that its
// global stores are coalesced 16-byte pieces; for this forward kernel the same staging plus scalar weight loads was
// built and measured slower (324 vs 236 us at 8 images) than lane-per-pixel reads with the weights broadcast from LDS.
constexpr int SD_TILE = 256;

__device__ __forceinline__ int64_t sd_padded_offset(int p, int H, int W, int U) {
  const int row = p / W;
  const int x = p - row * W, n = row / H, yy = row - n * H;
  return (((int64_t)n * (H + 2) + yy + 1) * (W + 2) + x + 1) * U;
}

template <int CM>
__global__ __launch_bounds__(256) void score_dense_kernel(const __bf16* __restrict__ u, const float* __restrict__ ws,
                                                         const float* __restrict__ bs, float* __restrict__ score, int N,
                                                         int H, int W, int U, int C) {
  extern __shared__ float wsm[];  // [U][CM]
  for (int i = threadIdx.x; i < U * CM; i += 256) {
    const int uu = i / CM, c = i - uu * CM;
    wsm[i] = c < C ? ws[uu * C + c] : 0.f;
  }
  __syncthreads();
  const int64_t npix = (int64_t)N * H * W;
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < npix; p += (int64_t)gridDim.x * 256) {
    const int x = (int)(p % W);
    const int64_t r = p / W;
    const int yy = (int)(r % H), n = (int)(r / H);
    const __bf16* src = u + (((int64_t)n * (H + 2) + yy + 1) * (W + 2) + x + 1) * U;
    float acc[CM];
#pragma unroll
    for (int c = 0; c < CM; ++c) acc[c] = c < C ? bs[c] : 0.f;
    for (int u0 = 0; u0 < U; u0 += 8) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(src + u0);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float f = (e & 1) ? bf_hi(v[e >> 1]) : bf_lo(v[e >> 1]);
        const float* wr = wsm + (u0 + e) * CM;
#pragma unroll
        for (int c = 0; c < CM; ++c) acc[c] = fmaf(f, wr[c], acc[c]);
      }
    }
#pragma unroll
    for (int c = 0; c < CM; ++c)
      if (c < C) score[p * C + c] = acc[c];
  }
}

// loss += -sum_pix log_softmax(logits)[label] / count; dlogits = (softmax - onehot) / count (0 for label < 0)
template <int CM>
__global__ __launch_bounds__(256) void softmax_ce_dense_kernel(const float* __restrict__ logits,
                                                              const int32_t* __restrict__ labels,
                                                              const unsigned long long* __restrict__ count, int C,
                                                              int64_t npix, double* __restrict__ loss,
                                                              float* __restrict__ dlogits,
                                                              const float* __restrict__ scale = nullptr,
                                                              const float* __restrict__ shift = nullptr,
                                                              double* __restrict__ partials = nullptr) {
  const float inv = 1.f / (1e-20f + (float)count[0]);
  // scale != nullptr: `logits` holds the raw scores and the batch norm's affine (xv_bn_dense_apply's expression) is applied
  // here, so the normalised scores never go to HBM
  float sc[CM], sh[CM];
#pragma unroll
  for (int c = 0; c < CM; ++c) sc[c] = (scale && c < C) ? scale[c] : 1.f, sh[c] = (scale && c < C) ? shift[c] : 0.f;
  double local = 0.0;
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < npix; p += (int64_t)gridDim.x * 256) {
    const int lab = labels[p];
    // one pass over the row, kept in registers (a lane's C floats are contiguous: the wave reads a dense span)
    // (C == CM, a multiple of four: the row as 16-byte loads / stores; one exponential per class on the transcendental
    // unit, the softmax as e / sum -- the scalar form spent 24 library exponentials and 24 four-byte accesses per pixel and ran
    // at 1.7 TB/s)
    float l[CM];
    const bool vec = C == CM && (CM & 3) == 0;
    if (vec) {
#pragma unroll
      for (int c4 = 0; c4 < CM / 4; ++c4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(logits + p * C + 4 * c4);
        l[4 * c4] = v.x, l[4 * c4 + 1] = v.y, l[4 * c4 + 2] = v.z, l[4 * c4 + 3] = v.w;
      }
      if (scale) {
#pragma unroll
        for (int c = 0; c < CM; ++c) l[c] = l[c] * sc[c] + sh[c];
      }
    } else {
#pragma unroll
      for (int c = 0; c < CM; ++c) l[c] = c < C ? (scale ? logits[p * C + c] * sc[c] + sh[c] : logits[p * C + c]) : -3.0e38f;
    }
    float m = l[0];
#pragma unroll
    for (int c = 1; c < CM; ++c) m = fmaxf(m, l[c]);
    float e[CM], sum = 0.f;
#pragma unroll
    for (int c = 0; c < CM; ++c) {
      e[c] = c < C ? xv_fast_exp(l[c] - m) : 0.f;
      sum += e[c];
    }
    const float lse = m + xv_fast_log(sum);
    const bool valid = lab >= 0 && lab < C;
    float llab = 0.f;
#pragma unroll
    for (int c = 0; c < CM; ++c)
      if (c == lab) llab = l[c];
    if (valid) local += (double)(lse - llab) * inv;
    const float k = valid ? inv / sum : 0.f, hot = valid ? inv : 0.f;
    float d[CM];
#pragma unroll
    for (int c = 0; c < CM; ++c) d[c] = e[c] * k - (c == lab ? hot : 0.f);
    if (vec) {
#pragma unroll
      for (int c4 = 0; c4 < CM / 4; ++c4)
        *reinterpret_cast<f32x4*>(dlogits + p * C + 4 * c4) = f32x4{d[4 * c4], d[4 * c4 + 1], d[4 * c4 + 2], d[4 * c4 + 3]};
    } else {
#pragma unroll
      for (int c = 0; c < CM; ++c)
        if (c < C) dlogits[p * C + c] = d[c];
    }
  }
  __shared__ double red[256];
  red[threadIdx.x] = local;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  // partials: one slot per workgroup, added up in workgroup order by loss_partials_sum_kernel (a bitwise reproducible
  // loss); without a workspace the workgroups add to the loss in arrival order (equal to ~1e-16 relative, not bit for bit)
  if (threadIdx.x == 0) {
    if (partials != nullptr)
      partials[blockIdx.x] = red[0];
    else
      atomicAdd(loss, red[0]);
  }
}

// loss += sum_b part[b] in a fixed order: thread t adds slots t, t + 256, ... in sequence, then the 256 sums meet in a tree
__global__ __launch_bounds__(256) void loss_partials_sum_kernel(const double* __restrict__ part, int n, double* __restrict__ loss) {
  double a = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) a += part[i];
  __shared__ double red[256];
  red[threadIdx.x] = a;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) *loss += red[0];
}

// dws[u][c] += sum_pix u[pix][u] * ds[pix][c]; dbs[c] += sum_pix ds[pix][c].  A block walks a contiguous run of pixels
// in tiles of 256 staged through LDS (coalesced 16-byte loads).  Thread (uu = t % 64, q = t / 64) owns channel uu for
// ALL classes and every fourth pixel of a tile: per pixel one 2-byte column read and CM/4 broadcast 16-byte reads feed
// CM FMAs.  (With the classes split over the four waves instead, every wave walked every pixel and the LDS pipe saw
// 4x the instructions: 503 us for 8 images against 0.4 GB of operands.)  The four partial sums meet in LDS before the
// global atomics.
template <int CM>
__global__ __launch_bounds__(256) void score_dense_wgrad_kernel(const __bf16* __restrict__ u, const float* __restrict__ ds,
                                                               float* __restrict__ dws, float* __restrict__ dbs, int N,
                                                               int H, int W, int U, int C, int64_t per_block) {
  constexpr int TP = 256;
  __shared__ __attribute__((aligned(16))) __bf16 ut[TP][64 + 8];  // +8: rows 144 B apart, conflict-free column reads
  __shared__ __attribute__((aligned(16))) float dt[TP][CM];
  __shared__ float part[3][64][CM + 1];
  const int uu = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int64_t npix = (int64_t)N * H * W;
  const int64_t p0 = (int64_t)blockIdx.x * per_block, p1 = p0 + per_block < npix ? p0 + per_block : npix;
  for (int ub = 0; ub < U; ub += 64) {
    float acc[CM], bacc[CM];
#pragma unroll
    for (int k = 0; k < CM; ++k) acc[k] = bacc[k] = 0.f;
    for (int64_t t0 = p0; t0 < p1; t0 += TP) {
      __syncthreads();
      for (int i = threadIdx.x; i < TP * 8; i += 256) {
        const int pp = i >> 3, piece = i & 7;
        const int64_t p = t0 + pp;
        u32x4 v = u32x4{0u, 0u, 0u, 0u};
        if (p < p1) {
          const int p32 = (int)p;  // npix < 2^31 (checked by the launcher)
          const int row = p32 / W;
          const int x = p32 - row * W, n = row / H, yy = row - n * H;
          v = *reinterpret_cast<const u32x4*>(u + (((int64_t)n * (H + 2) + yy + 1) * (W + 2) + x + 1) * U + ub + piece * 8);
        }
        *reinterpret_cast<u32x4*>(&ut[pp][piece * 8]) = v;
      }
      for (int i = threadIdx.x; i < TP * CM; i += 256) {
        const int pp = i / CM, c = i - pp * CM;
        const int64_t p = t0 + pp;
        dt[pp][c] = (p < p1 && c < C) ? ds[p * C + c] : 0.f;
      }
      __syncthreads();
#pragma unroll 4
      for (int pp = q; pp < TP; pp += 4) {
        const float f = (float)ut[pp][uu];
#pragma unroll
        for (int k4 = 0; k4 < CM; k4 += 4) {
          const f32x4 d = *reinterpret_cast<const f32x4*>(&dt[pp][k4]);  // wave-uniform address: broadcast
          acc[k4] = fmaf(f, d.x, acc[k4]);
          acc[k4 + 1] = fmaf(f, d.y, acc[k4 + 1]);
          acc[k4 + 2] = fmaf(f, d.z, acc[k4 + 2]);
          acc[k4 + 3] = fmaf(f, d.w, acc[k4 + 3]);
          bacc[k4] += d.x;
          bacc[k4 + 1] += d.y;
          bacc[k4 + 2] += d.z;
          bacc[k4 + 3] += d.w;
        }
      }
    }
    __syncthreads();
    if (q > 0) {
#pragma unroll
      for (int k = 0; k < CM; ++k) part[q - 1][uu][k] = acc[k];
      if (uu == 0) {
#pragma unroll
        for (int k = 0; k < CM; ++k) dt[q][k] = bacc[k];  // dt is free now: rows 1..3 carry the bias sums
      }
    }
    __syncthreads();
    if (q == 0) {
#pragma unroll
      for (int k = 0; k < CM; ++k) {
        if (k < C) {
          const float a = acc[k] + part[0][uu][k] + part[1][uu][k] + part[2][uu][k];
          atomicAdd(dws + (ub + uu) * C + k, a);
          if (ub == 0 && uu == 0) atomicAdd(dbs + k, bacc[k] + dt[1][k] + dt[2][k] + dt[3][k]);
        }
      }
    }
  }
}

// du[pix][u] = sum_c ds[pix][c] * ws[u][c]: thread per pixel (its C gradients in registers, the weights wave-uniform
// scalar loads), the U results go through LDS so that the global stores are coalesced 16-byte pieces (see
// score_dense_kernel)
template <int CM>
__global__ __launch_bounds__(256) void score_dense_dgrad_kernel(const float* __restrict__ ds, const float* __restrict__ ws,
                                                               __bf16* __restrict__ du, int N, int H, int W, int U,
                                                               int C) {
  extern __shared__ __attribute__((aligned(16))) __bf16 tile[];  // [256][U + 8]
  const int c8 = U >> 3, rs = U + 8;
  const int npix = N * H * W;
  for (int base = blockIdx.x * SD_TILE; base < npix; base += gridDim.x * SD_TILE) {
    const int p = base + threadIdx.x;
    float d[CM];
#pragma unroll
    for (int c = 0; c < CM; ++c) d[c] = (c < C && p < npix) ? ds[(int64_t)p * C + c] : 0.f;
    __syncthreads();
    for (int u0 = 0; u0 < U; u0 += 8) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float* wr = ws + (u0 + e) * C;  // wave-uniform
        float a = 0.f;
#pragma unroll
        for (int c = 0; c < CM; ++c)
          if (c < C) a = fmaf(d[c], wr[c], a);
        v[e] = a;
      }
      *reinterpret_cast<u32x4*>(tile + threadIdx.x * rs + u0) =
          u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
    }
    __syncthreads();
    for (int i = threadIdx.x; i < SD_TILE * c8; i += 256) {
      const int pp = i / c8, piece = i - pp * c8;
      if (base + pp < npix)
        *reinterpret_cast<u32x4*>(du + sd_padded_offset(base + pp, H, W, U) + piece * 8) =
            *reinterpret_cast<const u32x4*>(tile + pp * rs + piece * 8);
    }
  }
}

// ---- the per-pixel score conv and its data gradient on the matrix cores, straight from global memory ----------------
// One lane per pixel read its 128-byte row in eight uncoalesced 16-byte loads (and the data gradient fetched 96 weights per
// channel group through the scalar cache): 460 / 540 us for 0.8 GB each at 16 images.  As MFMA operands the same bytes are
// 64 contiguous bytes per pixel and instruction.
//   forward  score[px][c] = b[c] + sum_u y[px][u] W[u][c]: D[class][pixel] on v_mfma_f32_16x16x32_bf16, A = W^T (classes
//            padded to 16) split EXACTLY into three bf16 terms (fp32 weights: 8 + 8 + 8 significant bits), B = the pixels'
//            bf16 channels; every product is exact in fp32, only the order of the additions differs from an FMA chain.
//            Lane (pixel l & 15, group l >> 4) ends with classes 4g .. 4g + 3 of its pixel.
//   gradient du[px][u] = sum_c ds[px][c] W[u][c] in exact fp32 on v_mfma_f32_16x16x4_f32 (K = 4 classes per step).
typedef __attribute__((ext_vector_type(4))) float sd_f32x4;

__device__ __forceinline__ void sd_split3(const float (&v)[8], bf16x8& h, bf16x8& m, bf16x8& l) {
  uint32_t hh[8], mm[8], ll[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    hh[e] = __builtin_bit_cast(uint32_t, v[e]) & 0xffff0000u;
    const float r1 = v[e] - __builtin_bit_cast(float, hh[e]);  // exact
    mm[e] = __builtin_bit_cast(uint32_t, r1) & 0xffff0000u;
    ll[e] = __builtin_bit_cast(uint32_t, r1 - __builtin_bit_cast(float, mm[e]));  // exact, fits 8 bits
  }
  auto pk = [](const uint32_t (&a)[8]) {
    return __builtin_bit_cast(bf16x8, u32x4{(a[0] >> 16) | (a[1] & 0xffff0000u), (a[2] >> 16) | (a[3] & 0xffff0000u),
                                            (a[4] >> 16) | (a[5] & 0xffff0000u), (a[6] >> 16) | (a[7] & 0xffff0000u)});
  };
  h = pk(hh), m = pk(mm), l = pk(ll);
}

template <int KS>  // U / 32 K-steps
__global__ __launch_bounds__(256) void score_dense_mfma_kernel(const __bf16* __restrict__ u, const float* __restrict__ ws,
                                                              const float* __restrict__ bs, float* __restrict__ score, int N,
                                                              int H, int W, int C) {
  constexpr int U = 32 * KS;
  const int lane = threadIdx.x & 63, l15 = lane & 15, lg = lane >> 4;
  bf16x8 wh[KS], wm[KS], wl[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    float wv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) wv[j] = l15 < C ? ws[(32 * s + 8 * lg + j) * C + l15] : 0.f;
    sd_split3(wv, wh[s], wm[s], wl[s]);
  }
  sd_f32x4 bias;
#pragma unroll
  for (int r = 0; r < 4; ++r) bias[r] = 4 * lg + r < C ? bs[4 * lg + r] : 0.f;
  const int npix = N * H * W;  // < 2^31 (checked by the launcher)
  const int ngroups = (npix + 15) >> 4;
  const int wid = (int)(blockIdx.x * 4 + (threadIdx.x >> 6)), nw = (int)gridDim.x * 4;
  for (int g0 = wid; g0 < ngroups; g0 += 4 * nw) {
    u32x4 xf[4][KS];
    int pp[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {  // four groups' loads in flight
      const int p = (g0 + q * nw) * 16 + l15;
      pp[q] = (g0 + q * nw < ngroups && p < npix) ? p : -1;
      const __bf16* src = u + sd_padded_offset(pp[q] >= 0 ? p : 0, H, W, U) + 8 * lg;
#pragma unroll
      for (int s = 0; s < KS; ++s) xf[q][s] = *reinterpret_cast<const u32x4*>(src + 32 * s);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      sd_f32x4 acc = bias;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const bf16x8 x = __builtin_bit_cast(bf16x8, xf[q][s]);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[s], x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm[s], x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[s], x, acc, 0, 0, 0);
      }
      if (pp[q] >= 0) {
        float* dst = score + (int64_t)pp[q] * C + 4 * lg;
        if ((C & 3) == 0) {
          if (4 * lg < C) *reinterpret_cast<sd_f32x4*>(dst) = acc;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (4 * lg + r < C) dst[r] = acc[r];
        }
      }
    }
  }
}

// The forward apply pass of the batch norm behind the x8 deconv FUSED with the score conv: a lane's B fragment (8 channels of
// its pixel) is computed -- upsample_words of the low-resolution map, the batch norm's affine, relu, bf16 -- instead of loaded;
// it is stored to y (the map the backward pass reads: 16 pixels x 64-byte runs per instruction) and multiplied as in
// score_dense_mfma_kernel.  Replaces bn_ups8_apply8_kernel<false> + score_dense_mfma_kernel: one 0.6 GB store instead of a
// store and a load.  64 channels (KS = 2).
__global__ __launch_bounds__(256) void score_dense_ups8_mfma_kernel(const __bf16* __restrict__ low, const float* __restrict__ zsc,
                                                                   const float* __restrict__ zsh, const float* __restrict__ ws,
                                                                   const float* __restrict__ bs, __bf16* __restrict__ y,
                                                                   float* __restrict__ score, int N, int H, int W, int C) {
  constexpr int KS = 2, U = 64;
  const int lane = threadIdx.x & 63, l15 = lane & 15, lg = lane >> 4;
  bf16x8 wh[KS], wm[KS], wl[KS];
  float sc[KS][8], sh[KS][8];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    float wv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      wv[j] = l15 < C ? ws[(32 * s + 8 * lg + j) * C + l15] : 0.f;
      sc[s][j] = zsc[32 * s + 8 * lg + j], sh[s][j] = zsh[32 * s + 8 * lg + j];
    }
    sd_split3(wv, wh[s], wm[s], wl[s]);
  }
  sd_f32x4 bias;
#pragma unroll
  for (int r = 0; r < 4; ++r) bias[r] = 4 * lg + r < C ? bs[4 * lg + r] : 0.f;
  const int npix = N * H * W;  // < 2^31 (checked by the launcher)
  const int ngroups = (npix + 15) >> 4;
  const int Hi = H >> 3, Wi = W >> 3;
  const int wid = (int)(blockIdx.x * 4 + (threadIdx.x >> 6)), nw = (int)gridDim.x * 4;
  for (int g0 = wid; g0 < ngroups; g0 += 2 * nw) {
    u32x4 xf[2][KS];
    int pp[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int p = (g0 + q * nw) * 16 + l15;
      pp[q] = (g0 + q * nw < ngroups && p < npix) ? p : -1;
      const int pc = pp[q] >= 0 ? p : 0;
      const int row = pc / W, ox = pc - row * W, n = row / H, oy = row - n * H;
      __bf16* dst = y + (((int64_t)n * (H + 2) + oy + 1) * (W + 2) + ox + 1) * U + 8 * lg;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const u32x4 zv = upsample_words<8>(low, n, oy, ox, 4 * s + lg, Hi, Wi, U);
        u32x4 o;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          float a = fmaf(bf_lo(zv[w]), sc[s][2 * w], sh[s][2 * w]);
          float b = fmaf(bf_hi(zv[w]), sc[s][2 * w + 1], sh[s][2 * w + 1]);
          a = a > 0.f ? a : 0.f;
          b = b > 0.f ? b : 0.f;
          o[w] = pack_bf16x2(a, b);
        }
        xf[q][s] = o;
        if (pp[q] >= 0) *reinterpret_cast<u32x4*>(dst + 32 * s) = o;
      }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      sd_f32x4 acc = bias;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const bf16x8 x = __builtin_bit_cast(bf16x8, xf[q][s]);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[s], x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm[s], x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[s], x, acc, 0, 0, 0);
      }
      if (pp[q] >= 0) {
        float* dst = score + (int64_t)pp[q] * C + 4 * lg;
        if ((C & 3) == 0) {
          if (4 * lg < C) *reinterpret_cast<sd_f32x4*>(dst) = acc;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (4 * lg + r < C) dst[r] = acc[r];
        }
      }
    }
  }
}

template <int UB, int KC>  // U / 16 channel blocks, ceil(C / 4) class steps
__global__ __launch_bounds__(256) void score_dense_dgrad_mfma_kernel(const float* __restrict__ ds, const float* __restrict__ ws,
                                                                    __bf16* __restrict__ du, int N, int H, int W, int C) {
  constexpr int U = 16 * UB;
  __shared__ __attribute__((aligned(16))) uint32_t stage[4][512];  // 2 KB per wave: the store stage below
  const int lane = threadIdx.x & 63, l15 = lane & 15, lg = lane >> 4;
  float wa[UB][KC];
#pragma unroll
  for (int b = 0; b < UB; ++b)
#pragma unroll
    for (int s = 0; s < KC; ++s) wa[b][s] = 4 * s + lg < C ? ws[(16 * b + l15) * C + 4 * s + lg] : 0.f;
  const int npix = N * H * W;
  const int ngroups = (npix + 15) >> 4;
  const int wid = (int)(blockIdx.x * 4 + (threadIdx.x >> 6)), nw = (int)gridDim.x * 4;
  for (int g0 = wid; g0 < ngroups; g0 += 2 * nw) {
    float dv[2][KC];
    int pp[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int p = (g0 + q * nw) * 16 + l15;
      pp[q] = (g0 + q * nw < ngroups && p < npix) ? p : -1;
#pragma unroll
      for (int s = 0; s < KC; ++s) dv[q][s] = (pp[q] >= 0 && 4 * s + lg < C) ? ds[(int64_t)p * C + 4 * s + lg] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      // A lane holds 4 channels (8 bytes) of pixel l15 per channel block: stored directly, an instruction scattered 64 8-byte
      // pieces over 16 lines, and a pixel's 128-byte line was completed by four different instructions (279 us for a 0.6 GB
      // map).  Through 2 KB of LDS per wave instead: [pixel][128 B], read back as 16-byte pieces with four consecutive lanes on
      // one 64-byte run -- two store instructions of 16 x 64 bytes per group of 16 pixels.
      if constexpr (UB == 4) {
        char* const st = reinterpret_cast<char*>(stage[threadIdx.x >> 6]);
#pragma unroll
        for (int b = 0; b < UB; ++b) {
          sd_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s = 0; s < KC; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[b][s], dv[q][s], acc, 0, 0, 0);
          *reinterpret_cast<u32x2*>(st + l15 * 128 + b * 32 + lg * 8) = u32x2{pack_bf16x2(acc[0], acc[1]), pack_bf16x2(acc[2], acc[3])};
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int pq = (g0 + q * nw) * 16 + (lane >> 2);  // the pixel this lane stores
        const bool on = g0 + q * nw < ngroups && pq < npix;
        char* const dst = reinterpret_cast<char*>(du + sd_padded_offset(on ? pq : 0, H, W, U)) + (lane & 3) * 16;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const u32x4 r = *reinterpret_cast<const u32x4*>(st + (lane >> 2) * 128 + half * 64 + (lane & 3) * 16);
          if (on) *reinterpret_cast<u32x4*>(dst + half * 64) = r;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      } else {
        __bf16* dst = du + sd_padded_offset(pp[q] >= 0 ? pp[q] : 0, H, W, U) + 4 * lg;
#pragma unroll
        for (int b = 0; b < UB; ++b) {
          sd_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s = 0; s < KC; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[b][s], dv[q][s], acc, 0, 0, 0);
          if (pp[q] >= 0)
            *reinterpret_cast<u32x2*>(dst + 16 * b) = u32x2{pack_bf16x2(acc[0], acc[1]), pack_bf16x2(acc[2], acc[3])};
        }
      }
    }
  }
}

//   filter gradient dW[u][c] = sum_px y[px][u] ds[px][c] (+ db[c] = sum_px ds[px][c]) on v_mfma_f32_16x16x4_f32: K = 4
//            pixels per step, A = ds^T (class l & 15, pixel l >> 4), B = the pixels' channels.  The instruction takes ONE
//            element per lane, but a lane may load 16 bytes: lane (n < 8, pixel g) loads channels 8n .. 8n + 7 of its pixel
//            (128 contiguous bytes per pixel) and MFMA e = 0 .. 7 takes element e of every lane, i.e. column n of MFMA e is
//            channel 8n + e (columns 8 .. 15 idle: the matrix pipe is not the limit here).  A wave keeps the 64 x 16 block
//            of dW in 32 registers over its run of pixels; the four waves meet in LDS in wave order, one atomic per cell.
//            (A first version with one 2-byte load per lane and MFMA ran at 900 us against the FMA kernel's 640.)
template <int UE>  // U / 8 = channels per lane
__global__ __launch_bounds__(256) void score_dense_wgrad_mfma_kernel(const __bf16* __restrict__ u, const float* __restrict__ ds,
                                                                    float* __restrict__ dws, float* __restrict__ dbs, int N,
                                                                    int H, int W, int C, int quads_per_wave,
                                                                    float* __restrict__ part) {
  static_assert(UE == 8, "64 channels: 8 lanes x 8 channels per pixel");
  constexpr int U = 8 * UE;
  __shared__ float red[4][UE * 4 + 1][64];
  const int lane = threadIdx.x & 63, l15 = lane & 15, lg = lane >> 4, wave = threadIdx.x >> 6;
  const int npix = N * H * W;
  const int nquads = (npix + 3) >> 2;
  const int wid = (int)blockIdx.x * 4 + wave;
  const int q0 = wid * quads_per_wave, q1 = q0 + quads_per_wave < nquads ? q0 + quads_per_wave : nquads;
  sd_f32x4 acc[UE];
#pragma unroll
  for (int e = 0; e < UE; ++e) acc[e] = sd_f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;
  for (int q = q0; q < q1; q += 4) {
    u32x4 yv[4];
    float dv[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {  // four quads' loads in flight
      const int p = (q + t) * 4 + lg;
      const bool ok = q + t < q1 && p < npix;
      yv[t] = (ok && l15 < 8) ? *reinterpret_cast<const u32x4*>(u + sd_padded_offset(p, H, W, U) + 8 * l15) : u32x4{0u, 0u, 0u, 0u};
      dv[t] = (ok && l15 < C) ? ds[(int64_t)p * C + l15] : 0.f;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int e = 0; e < UE; ++e) {
        const float ye = (e & 1) ? bf_hi(yv[t][e >> 1]) : bf_lo(yv[t][e >> 1]);
        acc[e] = __builtin_amdgcn_mfma_f32_16x16x4f32(dv[t], ye, acc[e], 0, 0, 0);
      }
      bsum += dv[t];
    }
  }
  // lane (column n = l15, group lg): acc[e][r] = dW[channel 8 n + e][class 4 lg + r] (n < 8); bsum = a quarter of db[l15]
#pragma unroll
  for (int e = 0; e < UE; ++e)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][e * 4 + r][lane] = acc[e][r];
  red[wave][UE * 4][lane] = bsum;
  __syncthreads();
  for (int i = threadIdx.x; i < (UE * 4 + 1) * 64; i += 256) {
    const int row = i >> 6, ln = i & 63;
    const float v = ((red[0][row][ln] + red[1][row][ln]) + red[2][row][ln]) + red[3][row][ln];
    if (part != nullptr) {  // the workgroup's own row of partial sums: bn_sums_kernel + score_dense_wgrad_scatter_kernel add
      part[(int64_t)blockIdx.x * ((UE * 4 + 1) * 64) + i] = v;  // them in a fixed order (bitwise reproducible gradients)
      continue;
    }
    const int n = ln & 15, g = ln >> 4;
    if (row < UE * 4) {
      const int cls = 4 * g + (row & 3), ch = 8 * n + (row >> 2);
      if (n < 8 && cls < C && v != 0.f) atomicAdd(dws + ch * C + cls, v);
    } else if (n < C && v != 0.f) {
      atomicAdd(dbs + n, v);  // (the four pixel slots of class n add up here)
    }
  }
}

// the cells of score_dense_wgrad_mfma_kernel's partial rows, summed over the workgroups (sums: [33][64] doubles), onto the
// gradients: cell (row, lane) = dW[channel 8 (lane & 15) + row / 4][class 4 (lane >> 4) + row % 4]; row 32 = the four pixel
// slots of db[lane & 15], added in slot order
__global__ __launch_bounds__(256) void score_dense_wgrad_scatter_kernel(const double* __restrict__ sums, int C,
                                                                       float* __restrict__ dws, float* __restrict__ dbs) {
  for (int i = threadIdx.x; i < 32 * 64; i += 256) {
    const int row = i >> 6, ln = i & 63, n = ln & 15, g = ln >> 4;
    const int cls = 4 * g + (row & 3), ch = 8 * n + (row >> 2);
    if (n < 8 && cls < C) dws[ch * C + cls] += (float)sums[i];
  }
  if ((int)threadIdx.x < C) {
    const double* b = sums + 32 * 64 + threadIdx.x;
    dbs[threadIdx.x] += (float)(((b[0] + b[16]) + b[32]) + b[48]);
  }
}

// ... and of score_dense_wgrad_split_kernel's rows (conv_wgrad.hip): sums [17][64] doubles, row c < 16 = class c x 64 units,
// row 16 = db in its first 16 entries
__global__ __launch_bounds__(256) void score_dense_wgrad_split_scatter_kernel(const double* __restrict__ sums, int C,
                                                                             float* __restrict__ dws, float* __restrict__ dbs) {
  for (int i = threadIdx.x; i < C * 64; i += 256) {
    const int c = i >> 6, u = i & 63;
    dws[u * C + c] += (float)sums[i];
  }
  if ((int)threadIdx.x < C) dbs[threadIdx.x] += (float)sums[16 * 64 + threadIdx.x];
}

bool same_shape(const xv_act* a, const xv_act* b) { return a->n == b->n && a->h == b->h && a->w == b->w && a->c == b->c; }

}  // namespace

// The per-channel sums of a reduce launch of `grid` workgroups: with a workspace (xv_bn_workspace_bytes) every workgroup
// writes its partial sums to its own row and bn_sums_kernel adds the rows in a fixed tree -- bitwise reproducible
// statistics and gradients; without one: a memset and f64 atomics in arrival order.
namespace {
constexpr int BN_MAX_GRID = 2048;  // (512 left the reduce passes at 3.7 TB/s: 8 waves per CU with 4 loads in flight each)
// workgroups of a fixed-order reduce pass: enough to keep the loads of a large map in flight, but at least 16 elements per
// thread -- every workgroup is a row the small summing launch behind it has to add (37 of them per training step)
static int bn_reduce_grid(int64_t total) {
  int64_t g = (total + 256 * 16 - 1) / (256 * 16);
  if (g < 64) g = 64;
  const int cap = bn_grid(total, BN_MAX_GRID);
  return (int)(g < cap ? g : cap);
}
// dgamma / dbeta: a backward reduction -- the gradients are accumulated by the same launch that adds the rows (workspace
// form) or by bn_grads_kernel behind the atomics
template <class F>
int bn_sums_launch(double* sums, int len, int grid, void* ws, size_t ws_bytes, hipStream_t s, F launch,
                   float* dgamma = nullptr, float* dbeta = nullptr) {
  if (ws != nullptr) {
    if (ws_bytes < (size_t)BN_MAX_GRID * len * sizeof(float) || ((uintptr_t)ws & 15)) return XV_EWORKSPACE;
    launch((float*)ws);
    hipLaunchKernelGGL(bn_sums_kernel, dim3(len), dim3(256), 0, s, (const float*)ws, grid, len, sums, dgamma, dbeta);
    return XV_OK;
  }
  const hipError_t e = hipMemsetAsync(sums, 0, sizeof(double) * len, s);
  if (e != hipSuccess) return (int)e;
  launch((float*)nullptr);
  if (dgamma != nullptr) hipLaunchKernelGGL(bn_grads_kernel, dim3((len / 2 + 63) / 64), dim3(64), 0, s, sums, len / 2, dgamma, dbeta);
  return XV_OK;
}
}  // namespace

extern "C" size_t xv_bn_workspace_bytes(int channels) {
  return channels > 0 ? (size_t)BN_MAX_GRID * 2 * channels * sizeof(float) : 0;
}

// sums[0 .. len) = the column sums of `rows` ([nrows][len] floats: per-workgroup partial sums, e.g. of xv_conv2d_fwd_stats),
// added in a fixed tree
extern "C" int xv_bn_sums_from_rows(const float* rows, int nrows, int len, double* sums, void* stream) {
  XV_CHECK_ARG(rows && sums);
  XV_CHECK_SHAPE(nrows > 0 && len > 0 && len <= 65535);
  hipLaunchKernelGGL(bn_sums_kernel, dim3(len), dim3(256), 0, (hipStream_t)stream, rows, nrows, len, sums);
  return xv_launch_status();
}

extern "C" int xv_bn_stats_ws(const xv_act* z, double* sums, void* workspace, size_t workspace_bytes, void* stream) {
  XV_REQUIRE_BF16(z);
  XV_CHECK_ARG(z && z->data && sums);
  XV_CHECK_SHAPE(z->c >= 64 && 2048 % z->c == 0 && z->n > 0 && z->h > 0 && z->w > 0);  // C/8 divides the block size
  const int64_t total = (int64_t)z->n * z->h * z->w * (z->c >> 3);
  XV_CHECK_SHAPE(total < 0x7fff0000);
  hipStream_t s = (hipStream_t)stream;
  const int grid = bn_reduce_grid(total);
  const int rc = bn_sums_launch(sums, 2 * z->c, grid, workspace, workspace_bytes, s, [&](float* part) {
    hipLaunchKernelGGL(bn_reduce_kernel<0>, dim3(grid), dim3(256), 0, s, (const __bf16*)z->data, nullptr, nullptr, nullptr,
                       nullptr, sums, z->n, z->h, z->w, z->c, (const float*)nullptr, (const float*)nullptr, part);
  });
  return rc != XV_OK ? rc : xv_launch_status();
}
extern "C" int xv_bn_stats(const xv_act* z, double* sums, void* stream) { return xv_bn_stats_ws(z, sums, nullptr, 0, stream); }

// xv_bn_sums_from_rows + xv_bn_finalize in one launch (rows: [nrows][2 channels] per-workgroup partial sums); `sums` is left
// as xv_bn_sums_from_rows leaves it.  Single-process statistics only: a data-parallel run all-reduces `sums` between the two.
extern "C" int xv_bn_finalize_from_rows(const float* rows, int nrows, int channels, int64_t count, const float* gamma,
                                        const float* beta, float eps, float momentum, float* moving_mean, float* moving_var,
                                        float* mean, float* invstd, float* scale, float* shift, double* sums, void* stream) {
  XV_CHECK_ARG(rows && sums && gamma && beta && mean && invstd && scale && shift &&
               (moving_mean == nullptr) == (moving_var == nullptr));
  XV_CHECK_SHAPE(nrows > 0 && channels > 0 && channels <= 32767 && count > 0);
  hipLaunchKernelGGL(bn_sums_finalize_kernel, dim3(channels), dim3(256), 0, (hipStream_t)stream, rows, nrows, channels, sums,
                     (double)count, gamma, beta, eps, momentum, moving_mean, moving_var, mean, invstd, scale, shift);
  return xv_launch_status();
}

// xv_bn_stats_ws + xv_bn_finalize: the statistics pass and ONE launch for the row sums and the per-channel results
// (workspace required: the per-workgroup rows live there)
extern "C" int xv_bn_stats_finalize_ws(const xv_act* z, double* sums, void* workspace, size_t workspace_bytes,
                                       const float* gamma, const float* beta, float eps, float momentum, float* moving_mean,
                                       float* moving_var, float* mean, float* invstd, float* scale, float* shift,
                                       void* stream) {
  XV_REQUIRE_BF16(z);
  XV_CHECK_ARG(z && z->data && sums && workspace && gamma && beta && mean && invstd && scale && shift &&
               (moving_mean == nullptr) == (moving_var == nullptr));
  XV_CHECK_SHAPE(z->c >= 64 && 2048 % z->c == 0 && z->n > 0 && z->h > 0 && z->w > 0);
  const int64_t total = (int64_t)z->n * z->h * z->w * (z->c >> 3);
  XV_CHECK_SHAPE(total < 0x7fff0000);
  if (workspace_bytes < (size_t)BN_MAX_GRID * 2 * z->c * sizeof(float) || ((uintptr_t)workspace & 15)) return XV_EWORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const int grid = bn_reduce_grid(total);
  hipLaunchKernelGGL(bn_reduce_kernel<0>, dim3(grid), dim3(256), 0, s, (const __bf16*)z->data, nullptr, nullptr, nullptr, nullptr,
                     sums, z->n, z->h, z->w, z->c, (const float*)nullptr, (const float*)nullptr, (float*)workspace);
  hipLaunchKernelGGL(bn_sums_finalize_kernel, dim3(z->c), dim3(256), 0, s, (const float*)workspace, grid, z->c, sums,
                     (double)((int64_t)z->n * z->h * z->w), gamma, beta, eps, momentum, moving_mean, moving_var, mean, invstd,
                     scale, shift);
  return xv_launch_status();
}

extern "C" int xv_bn_finalize(const double* sums, int channels, int64_t count, const float* gamma, const float* beta,
                              float eps, float momentum, float* moving_mean, float* moving_var, float* mean,
                              float* invstd, float* scale, float* shift, void* stream) {
  XV_CHECK_ARG(sums && gamma && beta && mean && invstd && scale && shift && (moving_mean == nullptr) == (moving_var == nullptr));
  XV_CHECK_SHAPE(channels > 0 && count > 0);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((channels + 63) / 64), dim3(64), 0, (hipStream_t)stream, sums, channels,
                     (double)count, gamma, beta, eps, momentum, moving_mean, moving_var, mean, invstd, scale, shift);
  return xv_launch_status();
}

extern "C" int xv_bn_apply(const xv_act* z, const float* scale, const float* shift, int relu, const xv_act* y,
                           void* stream) {
  XV_REQUIRE_BF16(z, y);
  XV_CHECK_ARG(z && y && z->data && y->data && scale && shift);
  XV_CHECK_SHAPE(same_shape(z, y) && (z->c & 7) == 0);
  const int64_t total = (int64_t)z->n * z->h * z->w * (z->c >> 3);
  if (z->c >= 64 && 2048 % z->c == 0 && total < 0x7fff0000)  // C / 8 divides the block size: a thread keeps its channels
    hipLaunchKernelGGL(bn_apply_fast_kernel, dim3(bn_grid(total, 2048)), dim3(256), 0, (hipStream_t)stream,
                       (const __bf16*)z->data, scale, shift, relu, (__bf16*)y->data, z->n, z->h, z->w, z->c);
  else
    hipLaunchKernelGGL(bn_apply_kernel, dim3(bn_grid(total, 8192)), dim3(256), 0, (hipStream_t)stream,
                       (const __bf16*)z->data, scale, shift, relu, (__bf16*)y->data, z->n, z->h, z->w, z->c);
  return xv_launch_status();
}

// y = relu(z * scale + shift) and pooled = maxpool2x2(y) in one pass (y may be a NULL-data descriptor: only the pooled map)
extern "C" int xv_bn_apply_pool(const xv_act* z, const float* scale, const float* shift, const xv_act* y, const xv_act* pooled,
                                void* stream) {
  XV_REQUIRE_BF16(z, y, pooled);
  XV_CHECK_ARG(z && pooled && z->data && pooled->data && scale && shift);
  XV_CHECK_SHAPE(z->c >= 64 && 2048 % z->c == 0 && (z->h & 1) == 0 && (z->w & 1) == 0);
  XV_CHECK_SHAPE(pooled->n == z->n && pooled->h == z->h / 2 && pooled->w == z->w / 2 && pooled->c == z->c);
  __bf16* yp = nullptr;
  if (y && y->data) {
    XV_CHECK_SHAPE(same_shape(y, z));
    yp = (__bf16*)y->data;
  }
  const int64_t total = (int64_t)z->n * (z->h / 2) * (z->w / 2) * (z->c >> 3);
  XV_CHECK_SHAPE(total < 0x7fff0000);
  hipLaunchKernelGGL(bn_apply_pool_kernel, dim3(bn_grid(total, 2048)), dim3(256), 0, (hipStream_t)stream,
                     (const __bf16*)z->data, scale, shift, yp, (__bf16*)pooled->data, z->n, z->h / 2, z->w / 2, z->c);
  return xv_launch_status();
}

// The gradient in two steps, so that a data-parallel caller can all-reduce `sums` in between (Sync-BN): the reduce
// step also adds the LOCAL sums into dgamma / dbeta (the gradient all-reduce sums those over ranks), the apply step
// uses whatever `sums` / `count` hold by then (global under Sync-BN).
extern "C" int xv_bn_bwd_reduce_ws(const xv_act* dy, const xv_act* y, const xv_act* z, const float* mean,
                                   const float* invstd, double* sums, float* dgamma, float* dbeta, void* workspace,
                                   size_t workspace_bytes, void* stream) {
  XV_REQUIRE_BF16(dy, y, z);
  XV_CHECK_ARG(dy && z && dy->data && z->data && mean && invstd && sums && dgamma && dbeta);
  XV_CHECK_SHAPE(same_shape(dy, z) && z->c >= 64 && 2048 % z->c == 0);
  const __bf16* yp = nullptr;
  if (y && y->data) {
    XV_CHECK_SHAPE(same_shape(y, z));
    yp = (const __bf16*)y->data;
  }
  hipStream_t s = (hipStream_t)stream;
  const int64_t total = (int64_t)z->n * z->h * z->w * (z->c >> 3);
  XV_CHECK_SHAPE(total < 0x7fff0000);
  const int grid = bn_reduce_grid(total);
  const int rc = bn_sums_launch(sums, 2 * z->c, grid, workspace, workspace_bytes, s, [&](float* part) {
    hipLaunchKernelGGL(bn_reduce_kernel<1>, dim3(grid), dim3(256), 0, s, (const __bf16*)z->data, (const __bf16*)dy->data, yp,
                       mean, invstd, sums, z->n, z->h, z->w, z->c, (const float*)nullptr, (const float*)nullptr, part);
  }, dgamma, dbeta);
  if (rc != XV_OK) return rc;
  return xv_launch_status();
}
extern "C" int xv_bn_bwd_reduce(const xv_act* dy, const xv_act* y, const xv_act* z, const float* mean,
                                const float* invstd, double* sums, float* dgamma, float* dbeta, void* stream) {
  return xv_bn_bwd_reduce_ws(dy, y, z, mean, invstd, sums, dgamma, dbeta, nullptr, 0, stream);
}

extern "C" int xv_bn_bwd_apply(const xv_act* dy, const xv_act* y, const xv_act* z, const float* mean,
                               const float* invstd, const float* gamma, const double* sums, int64_t count,
                               const xv_act* dz, void* stream) {
  XV_REQUIRE_BF16(dy, y, z, dz);
  XV_CHECK_ARG(dy && z && dz && dy->data && z->data && dz->data && mean && invstd && gamma && sums);
  XV_CHECK_SHAPE(same_shape(dy, z) && same_shape(dz, z) && (z->c & 7) == 0 && count > 0);
  const __bf16* yp = (y && y->data) ? (const __bf16*)y->data : nullptr;
  const int64_t total = (int64_t)z->n * z->h * z->w * (z->c >> 3);
  if (z->c >= 64 && 2048 % z->c == 0 && total < 0x7fff0000)
    hipLaunchKernelGGL(bn_bwd_apply_fast_kernel<false>, dim3(bn_grid(total, 2048)), dim3(256), 0, (hipStream_t)stream,
                       (const __bf16*)dy->data, yp, (const __bf16*)z->data, mean, invstd, gamma, sums, (double)count,
                       (__bf16*)dz->data, z->n, z->h, z->w, z->c, (const float*)nullptr, (const float*)nullptr);
  else
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(bn_grid(total, 8192)), dim3(256), 0, (hipStream_t)stream,
                       (const __bf16*)dy->data, yp, (const __bf16*)z->data, mean, invstd, gamma, sums, (double)count,
                       (__bf16*)dz->data, z->n, z->h, z->w, z->c);
  return xv_launch_status();
}

// The same two steps with the relu mask recomputed from z -- relu(z * scale + shift) > 0 exactly where z * scale + shift > 0
// (the forward pass's own fp32 expression; a positive value never rounds to a bf16 zero above 2^-134) -- so the activation
// map is not read: a third less traffic in the reduce step, a quarter less in the apply step.  Trunk channel counts only.
extern "C" int xv_bn_bwd_reduce_zmask(const xv_act* dy, const xv_act* z, const float* mean, const float* invstd,
                                      const float* scale, const float* shift, double* sums, float* dgamma, float* dbeta,
                                      void* workspace, size_t workspace_bytes, void* stream) {
  XV_REQUIRE_BF16(dy, z);
  XV_CHECK_ARG(dy && z && dy->data && z->data && mean && invstd && scale && shift && sums && dgamma && dbeta);
  XV_CHECK_SHAPE(same_shape(dy, z) && z->c >= 64 && 2048 % z->c == 0);
  hipStream_t s = (hipStream_t)stream;
  const int64_t total = (int64_t)z->n * z->h * z->w * (z->c >> 3);
  XV_CHECK_SHAPE(total < 0x7fff0000);
  const int grid = bn_reduce_grid(total);
  const int rc = bn_sums_launch(sums, 2 * z->c, grid, workspace, workspace_bytes, s, [&](float* part) {
    hipLaunchKernelGGL(bn_reduce_kernel<2>, dim3(grid), dim3(256), 0, s, (const __bf16*)z->data, (const __bf16*)dy->data,
                       (const __bf16*)nullptr, mean, invstd, sums, z->n, z->h, z->w, z->c, scale, shift, part);
  }, dgamma, dbeta);
  if (rc != XV_OK) return rc;
  return xv_launch_status();
}

extern "C" int xv_bn_bwd_apply_zmask(const xv_act* dy, const xv_act* z, const float* mean, const float* invstd,
                                     const float* scale, const float* shift, const float* gamma, const double* sums,
                                     int64_t count, const xv_act* dz, void* stream) {
  XV_REQUIRE_BF16(dy, z, dz);
  XV_CHECK_ARG(dy && z && dz && dy->data && z->data && dz->data && mean && invstd && scale && shift && gamma && sums);
  XV_CHECK_SHAPE(same_shape(dy, z) && same_shape(dz, z) && z->c >= 64 && 2048 % z->c == 0 && count > 0);
  const int64_t total = (int64_t)z->n * z->h * z->w * (z->c >> 3);
  XV_CHECK_SHAPE(total < 0x7fff0000);
  hipLaunchKernelGGL(bn_bwd_apply_fast_kernel<true>, dim3(bn_grid(total, 2048)), dim3(256), 0, (hipStream_t)stream,
                     (const __bf16*)dy->data, (const __bf16*)nullptr, (const __bf16*)z->data, mean, invstd, gamma, sums,
                     (double)count, (__bf16*)dz->data, z->n, z->h, z->w, z->c, scale, shift);
  return xv_launch_status();
}

// ---- the batch norm BEHIND the x8 deconv without its input in memory ------------------------------------------------------
// `low` is the deconv's input (padded bf16, 64 .. 2048 channels); the normalised map is z = bilinear_x8(low) rounded to bf16
// -- exactly what xv_upsample_raw_fwd(low, 8) stores -- recomputed per element (four 16-byte reads of a map 64 times smaller,
// cache-resident) in the four passes that would read it back: statistics + finalize, apply, gradient sums, gradient.  The
// training step of `batch_normalization: true` then never writes or reads the 0.6 GB pre-activation of `upscore`
// (custom_layers.py:112-119 behind simple_fcn.py:117-119): five of its thirteen passes over maps of that size.
// grid of bn_ups8_reduce8_kernel: x over the row's (phase group, channel group) pairs, y over rows; x * y <= BN_MAX_GRID
static dim3 ups8_reduce_grid(int n, int h, int w, int c) {
  const unsigned gx = (unsigned)(((w / 8 + 1) * (c >> 3) + 255) / 256);
  unsigned gy = (unsigned)BN_MAX_GRID / (gx < 1 ? 1 : gx);
  if (gy < 1) gy = 1;
  if ((int64_t)gy > (int64_t)n * h) gy = (unsigned)(n * h);
  return dim3(gx, gy);
}
static bool ups_ok(const xv_act* low, int n, int h, int w, int c) {
  return low && low->data && low->dtype == XV_BF16 && low->n == n && 8 * low->h == h && 8 * low->w == w && low->c == c &&
         c >= 64 && 2048 % c == 0 && (int64_t)n * h * w * (c >> 3) < 0x7fff0000;
}
extern "C" int xv_bn_stats_finalize_ups8_ws(const xv_act* low, double* sums, void* workspace, size_t workspace_bytes,
                                            const float* gamma, const float* beta, float eps, float momentum, float* moving_mean,
                                            float* moving_var, float* mean, float* invstd, float* scale, float* shift,
                                            void* stream) {
  XV_CHECK_ARG(low && sums && workspace && gamma && beta && mean && invstd && scale && shift &&
               (moving_mean == nullptr) == (moving_var == nullptr));
  const int n = low->n, h = 8 * low->h, w = 8 * low->w, c = low->c;
  XV_CHECK_SHAPE(ups_ok(low, n, h, w, c));
  if (workspace_bytes < (size_t)BN_MAX_GRID * 2 * c * sizeof(float) || ((uintptr_t)workspace & 15)) return XV_EWORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const dim3 g2 = ups8_reduce_grid(n, h, w, c);
  const int grid = (int)(g2.x * g2.y);
  XV_CHECK_SHAPE(grid <= BN_MAX_GRID);
  hipLaunchKernelGGL(bn_ups8_reduce8_kernel<false>, g2, dim3(256), 0, s, (const __bf16*)low->data, (const __bf16*)nullptr,
                     (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (float*)workspace, n,
                     h, w, c);
  hipLaunchKernelGGL(bn_sums_finalize_kernel, dim3(c), dim3(256), 0, s, (const float*)workspace, grid, c, sums,
                     (double)((int64_t)n * h * w), gamma, beta, eps, momentum, moving_mean, moving_var, mean, invstd, scale, shift);
  return xv_launch_status();
}
// The statistics pass alone (sums[c] = sum z, sums[C + c] = sum z^2 over the recomputed map), for data-parallel runs: the
// caller all-reduces `sums` over the ranks and finalises with xv_bn_finalize (count = pixels of the GLOBAL batch).  The same
// per-workgroup rows in the same order as xv_bn_stats_finalize_ups8_ws: bitwise reproducible.
extern "C" int xv_bn_stats_ups8_ws(const xv_act* low, double* sums, void* workspace, size_t workspace_bytes, void* stream) {
  XV_CHECK_ARG(low && sums && workspace);
  const int n = low->n, h = 8 * low->h, w = 8 * low->w, c = low->c;
  XV_CHECK_SHAPE(ups_ok(low, n, h, w, c));
  if (workspace_bytes < (size_t)BN_MAX_GRID * 2 * c * sizeof(float) || ((uintptr_t)workspace & 15)) return XV_EWORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const dim3 g2 = ups8_reduce_grid(n, h, w, c);
  const int grid = (int)(g2.x * g2.y);
  XV_CHECK_SHAPE(grid <= BN_MAX_GRID);
  hipLaunchKernelGGL(bn_ups8_reduce8_kernel<false>, g2, dim3(256), 0, s, (const __bf16*)low->data, (const __bf16*)nullptr,
                     (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (float*)workspace, n,
                     h, w, c);
  hipLaunchKernelGGL(bn_sums_kernel, dim3(2 * c), dim3(256), 0, s, (const float*)workspace, grid, 2 * c, sums, (float*)nullptr,
                     (float*)nullptr);
  return xv_launch_status();
}
extern "C" int xv_bn_apply_ups8(const xv_act* low, const float* scale, const float* shift, int relu, const xv_act* y, void* stream) {
  XV_REQUIRE_BF16(y);
  XV_CHECK_ARG(low && y && y->data && scale && shift);
  XV_CHECK_SHAPE(ups_ok(low, y->n, y->h, y->w, y->c));
  const int rows = y->n * y->h;
  const dim3 grid((unsigned)(((y->w / 8 + 1) * (y->c >> 3) + 255) / 256), (unsigned)(rows < 512 ? rows : 512));
  hipLaunchKernelGGL(bn_ups8_apply8_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, (const __bf16*)low->data,
                     (const __bf16*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr,
                     (const double*)nullptr, 1.0, scale, shift, relu, (__bf16*)y->data, y->n, y->h, y->w, y->c);
  return xv_launch_status();
}
extern "C" int xv_bn_bwd_reduce_zmask_ups8(const xv_act* dy, const xv_act* low, const float* mean, const float* invstd,
                                           const float* scale, const float* shift, double* sums, float* dgamma, float* dbeta,
                                           void* workspace, size_t workspace_bytes, void* stream) {
  XV_REQUIRE_BF16(dy);
  XV_CHECK_ARG(dy && low && dy->data && mean && invstd && scale && shift && sums && dgamma && dbeta);
  XV_CHECK_SHAPE(ups_ok(low, dy->n, dy->h, dy->w, dy->c));
  hipStream_t s = (hipStream_t)stream;
  if (workspace == nullptr) return XV_EWORKSPACE;  // (the eight-pixel form has no atomic variant)
  const dim3 g2 = ups8_reduce_grid(dy->n, dy->h, dy->w, dy->c);
  const int grid = (int)(g2.x * g2.y);
  XV_CHECK_SHAPE(grid <= BN_MAX_GRID);
  const int rc = bn_sums_launch(sums, 2 * dy->c, grid, workspace, workspace_bytes, s, [&](float* part) {
    hipLaunchKernelGGL(bn_ups8_reduce8_kernel<true>, g2, dim3(256), 0, s, (const __bf16*)low->data, (const __bf16*)dy->data, mean,
                       invstd, scale, shift, part, dy->n, dy->h, dy->w, dy->c);
  }, dgamma, dbeta);
  if (rc != XV_OK) return rc;
  return xv_launch_status();
}
extern "C" int xv_bn_bwd_apply_zmask_ups8(const xv_act* dy, const xv_act* low, const float* mean, const float* invstd,
                                          const float* scale, const float* shift, const float* gamma, const double* sums,
                                          int64_t count, const xv_act* dz, void* stream) {
  XV_REQUIRE_BF16(dy, dz);
  XV_CHECK_ARG(dy && low && dz && dy->data && dz->data && mean && invstd && scale && shift && gamma && sums);
  XV_CHECK_SHAPE(same_shape(dy, dz) && ups_ok(low, dy->n, dy->h, dy->w, dy->c) && count > 0);
  const int rows = dy->n * dy->h;
  const dim3 grid((unsigned)(((dy->w / 8 + 1) * (dy->c >> 3) + 255) / 256), (unsigned)(rows < 512 ? rows : 512));
  hipLaunchKernelGGL(bn_ups8_apply8_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, (const __bf16*)low->data,
                     (const __bf16*)dy->data, mean, invstd, gamma, sums, (double)count, scale, shift, 1, (__bf16*)dz->data, dy->n,
                     dy->h, dy->w, dy->c);
  return xv_launch_status();
}

// The batch-norm gradient of a conv -> batch norm -> relu -> 2x2 max-pool block straight from the gradient of the POOLED
// map (see bn_pool_bwd_kernel): replaces xv_maxpool2x2_bwd + xv_bn_bwd_reduce + xv_bn_bwd_apply and the full-resolution
// routed-gradient map between them.  z: the conv output [n][h][w][c], dpooled: [n][h/2][w/2][c]; count = n h w (x ranks).
extern "C" int xv_bn_pool_bwd_reduce(const xv_act* dpooled, const xv_act* z, const float* mean, const float* invstd,
                                     const float* scale, const float* shift, double* sums, float* dgamma, float* dbeta,
                                     void* workspace, size_t workspace_bytes, void* stream) {
  XV_REQUIRE_BF16(dpooled, z);
  XV_CHECK_ARG(dpooled && z && dpooled->data && z->data && mean && invstd && scale && shift && sums && dgamma && dbeta);
  XV_CHECK_SHAPE(z->c >= 64 && 2048 % z->c == 0 && (z->h & 1) == 0 && (z->w & 1) == 0);
  XV_CHECK_SHAPE(dpooled->n == z->n && dpooled->h == z->h / 2 && dpooled->w == z->w / 2 && dpooled->c == z->c);
  hipStream_t s = (hipStream_t)stream;
  const int64_t total = (int64_t)z->n * (z->h / 2) * (z->w / 2) * (z->c >> 3);
  XV_CHECK_SHAPE(total < 0x7fff0000);
  const int grid = bn_reduce_grid(total);
  const int rc = bn_sums_launch(sums, 2 * z->c, grid, workspace, workspace_bytes, s, [&](float* part) {
    hipLaunchKernelGGL(bn_pool_bwd_kernel<0>, dim3(grid), dim3(256), 0, s, (const __bf16*)dpooled->data, (const __bf16*)z->data,
                       mean, invstd, scale, shift, (const float*)nullptr, sums, 1.0, (__bf16*)nullptr, z->n, z->h / 2, z->w / 2,
                       z->c, part);
  }, dgamma, dbeta);
  if (rc != XV_OK) return rc;
  return xv_launch_status();
}

extern "C" int xv_bn_pool_bwd_apply(const xv_act* dpooled, const xv_act* z, const float* mean, const float* invstd,
                                    const float* scale, const float* shift, const float* gamma, const double* sums,
                                    int64_t count, const xv_act* dz, void* stream) {
  XV_REQUIRE_BF16(dpooled, z, dz);
  XV_CHECK_ARG(dpooled && z && dz && dpooled->data && z->data && dz->data && mean && invstd && scale && shift && gamma && sums);
  XV_CHECK_SHAPE(same_shape(dz, z) && z->c >= 64 && 2048 % z->c == 0 && (z->h & 1) == 0 && (z->w & 1) == 0 && count > 0);
  XV_CHECK_SHAPE(dpooled->n == z->n && dpooled->h == z->h / 2 && dpooled->w == z->w / 2 && dpooled->c == z->c);
  const int64_t total = (int64_t)z->n * (z->h / 2) * (z->w / 2) * (z->c >> 3);
  XV_CHECK_SHAPE(total < 0x7fff0000);
  hipLaunchKernelGGL(bn_pool_bwd_kernel<1>, dim3(bn_grid(total, 2048)), dim3(256), 0, (hipStream_t)stream,
                     (const __bf16*)dpooled->data, (const __bf16*)z->data, mean, invstd, scale, shift, gamma,
                     const_cast<double*>(sums), (double)count, (__bf16*)dz->data, z->n, z->h / 2, z->w / 2, z->c, (float*)nullptr);
  return xv_launch_status();
}

extern "C" int xv_bn_bwd(const xv_act* dy, const xv_act* y, const xv_act* z, const float* mean, const float* invstd,
                         const float* gamma, double* sums, float* dgamma, float* dbeta, const xv_act* dz, void* stream) {
  XV_REQUIRE_BF16(dy, y, z, dz);
  XV_CHECK_ARG(dz && z);
  const int rc = xv_bn_bwd_reduce(dy, y, z, mean, invstd, sums, dgamma, dbeta, stream);
  if (rc != XV_OK) return rc;
  return xv_bn_bwd_apply(dy, y, z, mean, invstd, gamma, sums, (int64_t)z->n * z->h * z->w, dz, stream);
}

extern "C" int xv_bn_dense_stats_ws(const float* z, int64_t rows, int channels, double* sums, void* workspace,
                                    size_t workspace_bytes, void* stream) {
  XV_CHECK_ARG(z && sums);
  XV_CHECK_SHAPE(rows > 0 && channels >= 1 && channels <= 32);
  hipStream_t s = (hipStream_t)stream;
  const int grid = bn_grid(rows, 512);  // (a thread keeps 2 x 32 sums: amortised over many rows)
  const int rc = bn_sums_launch(sums, 2 * channels, grid, workspace, workspace_bytes, s, [&](float* part) {
    hipLaunchKernelGGL(bn_dense_reduce_kernel<0>, dim3(grid), dim3(256), 0, s, z, nullptr, nullptr, nullptr, sums, rows, channels,
                       part);
  });
  return rc != XV_OK ? rc : xv_launch_status();
}
extern "C" int xv_bn_dense_stats(const float* z, int64_t rows, int channels, double* sums, void* stream) {
  return xv_bn_dense_stats_ws(z, rows, channels, sums, nullptr, 0, stream);
}

extern "C" int xv_bn_dense_apply(const float* z, int64_t rows, int channels, const float* scale, const float* shift,
                                 float* y, void* stream) {
  XV_CHECK_ARG(z && y && scale && shift);
  XV_CHECK_SHAPE(rows > 0 && channels >= 1 && channels <= 32);
  hipLaunchKernelGGL(bn_dense_apply_kernel, dim3(bn_grid(rows * channels, 8192)), dim3(256), 0, (hipStream_t)stream, z,
                     scale, shift, y, rows * channels, channels);
  return xv_launch_status();
}

extern "C" int xv_bn_dense_bwd_reduce_ws(const float* dy, const float* z, int64_t rows, int channels, const float* mean,
                                         const float* invstd, double* sums, float* dgamma, float* dbeta, void* workspace,
                                         size_t workspace_bytes, void* stream) {
  XV_CHECK_ARG(dy && z && mean && invstd && sums && dgamma && dbeta);
  XV_CHECK_SHAPE(rows > 0 && channels >= 1 && channels <= 32);
  hipStream_t s = (hipStream_t)stream;
  const int grid = bn_grid(rows, 512);  // (a thread keeps 2 x 32 sums: amortised over many rows)
  const int rc = bn_sums_launch(sums, 2 * channels, grid, workspace, workspace_bytes, s, [&](float* part) {
    hipLaunchKernelGGL(bn_dense_reduce_kernel<1>, dim3(grid), dim3(256), 0, s, z, dy, mean, invstd, sums, rows, channels, part);
  }, dgamma, dbeta);
  if (rc != XV_OK) return rc;
  return xv_launch_status();
}
extern "C" int xv_bn_dense_bwd_reduce(const float* dy, const float* z, int64_t rows, int channels, const float* mean,
                                      const float* invstd, double* sums, float* dgamma, float* dbeta, void* stream) {
  return xv_bn_dense_bwd_reduce_ws(dy, z, rows, channels, mean, invstd, sums, dgamma, dbeta, nullptr, 0, stream);
}

extern "C" int xv_bn_dense_bwd_apply(const float* dy, const float* z, int64_t rows, int channels, const float* mean,
                                     const float* invstd, const float* gamma, const double* sums, int64_t count,
                                     float* dz, void* stream) {
  XV_CHECK_ARG(dy && z && mean && invstd && gamma && sums && dz);
  XV_CHECK_SHAPE(rows > 0 && channels >= 1 && channels <= 32 && count > 0);
  if ((channels == 4 || channels == 8 || channels == 12 || channels == 16) &&
      (((uintptr_t)dy | (uintptr_t)z | (uintptr_t)dz) & 15) == 0) {
    const dim3 grid(bn_grid(rows, 2048));
#define XV_BDA(CMV)                                                                                                          \
  hipLaunchKernelGGL(bn_dense_bwd_apply_rows_kernel<CMV>, grid, dim3(256), 0, (hipStream_t)stream, dy, z, mean, invstd, gamma, \
                     sums, (double)count, dz, rows)
    switch (channels) {
      case 4: XV_BDA(4); break;
      case 8: XV_BDA(8); break;
      case 12: XV_BDA(12); break;
      default: XV_BDA(16); break;
    }
#undef XV_BDA
    return xv_launch_status();
  }
  hipLaunchKernelGGL(bn_dense_bwd_apply_kernel, dim3(bn_grid(rows * channels, 8192)), dim3(256), 0, (hipStream_t)stream,
                     dy, z, mean, invstd, gamma, sums, (double)count, dz, rows * channels, channels);
  return xv_launch_status();
}

extern "C" int xv_bn_dense_bwd(const float* dy, const float* z, int64_t rows, int channels, const float* mean,
                               const float* invstd, const float* gamma, double* sums, float* dgamma, float* dbeta,
                               float* dz, void* stream) {
  const int rc = xv_bn_dense_bwd_reduce(dy, z, rows, channels, mean, invstd, sums, dgamma, dbeta, stream);
  if (rc != XV_OK) return rc;
  return xv_bn_dense_bwd_apply(dy, z, rows, channels, mean, invstd, gamma, sums, rows, dz, stream);
}

extern "C" int xv_upsample_raw_fwd(const xv_act* x, int factor, const xv_act* y, void* stream) {
  XV_REQUIRE_BF16(x, y);
  XV_CHECK_ARG(x && y && x->data && y->data);
  XV_CHECK_SHAPE((factor == 2 || factor == 8) && (x->c & 7) == 0 && y->n == x->n && y->h == factor * x->h &&
                 y->w == factor * x->w && y->c == x->c);
  XV_CHECK_SHAPE(y->h <= 65535 && y->n <= 65535 && (int64_t)y->w * (y->c >> 3) < 0x7fff0000);
  const dim3 grid((unsigned)(((int64_t)y->w * (y->c >> 3) + 255) / 256), (unsigned)y->h, (unsigned)y->n);
  if (factor == 2)
    hipLaunchKernelGGL(upsample_raw_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, (const __bf16*)x->data, (__bf16*)y->data,
                       x->n, x->h, x->w, x->c);
  else
    hipLaunchKernelGGL(upsample_raw_kernel<8>, grid, dim3(256), 0, (hipStream_t)stream, (const __bf16*)x->data, (__bf16*)y->data,
                       x->n, x->h, x->w, x->c);
  return xv_launch_status();
}

extern "C" int xv_upsample_raw_bwd(const xv_act* dy, int factor, const xv_act* dx, void* stream) {
  XV_REQUIRE_BF16(dy, dx);
  XV_CHECK_ARG(dx && dy && dx->data && dy->data);
  XV_CHECK_SHAPE((factor == 2 || factor == 8) && (dx->c & 7) == 0 && dy->n == dx->n && dy->h == factor * dx->h &&
                 dy->w == factor * dx->w && dy->c == dx->c);
  const int64_t total = (int64_t)dx->n * dx->h * dx->w * (dx->c >> 3);
  XV_CHECK_SHAPE(total < 0x7fff0000);  // (32-bit index arithmetic in the kernel)
  if (factor == 2)
    hipLaunchKernelGGL(upsample_raw_bwd_kernel<2>, dim3(bn_grid(total, 8192)), dim3(256), 0, (hipStream_t)stream,
                       (const __bf16*)dy->data, (__bf16*)dx->data, dx->n, dx->h, dx->w, dx->c);
  else
    hipLaunchKernelGGL(upsample_raw_bwd_kernel<8>, dim3(bn_grid(total, 8192)), dim3(256), 0, (hipStream_t)stream,
                       (const __bf16*)dy->data, (__bf16*)dx->data, dx->n, dx->h, dx->w, dx->c);
  return xv_launch_status();
}

// The x8 gradient in two passes over block sums (upsample8_bwd_blocks_kernel): every gradient element is read once.
// workspace: 4 (n) (h + 1) (w + 1) c floats for a low-resolution map [n][h][w][c].
extern "C" size_t xv_upsample_raw_bwd_workspace_bytes(int n, int h, int w, int c) {
  if (!xv_dims_sane(n, h, w) || c <= 0 || (c & 7)) return 0;
  return (size_t)4 * n * ((size_t)h + 1) * ((size_t)w + 1) * c * sizeof(float);
}
extern "C" int xv_upsample_raw_bwd_ws(const xv_act* dy, int factor, const xv_act* dx, void* workspace, size_t workspace_bytes,
                                      void* stream) {
  if (factor != 8 || workspace == nullptr) return xv_upsample_raw_bwd(dy, factor, dx, stream);
  XV_REQUIRE_BF16(dy, dx);
  XV_CHECK_ARG(dx && dy && dx->data && dy->data);
  XV_CHECK_SHAPE((dx->c & 7) == 0 && dy->n == dx->n && dy->h == 8 * dx->h && dy->w == 8 * dx->w && dy->c == dx->c);
  XV_CHECK_SHAPE(dx->h < 65535 && dx->n <= 65535 && (int64_t)(dx->w + 1) * (dx->c >> 3) < 0x7fff0000);
  if (workspace_bytes < xv_upsample_raw_bwd_workspace_bytes(dx->n, dx->h, dx->w, dx->c)) return XV_EWORKSPACE;
  XV_CHECK_ARG(((uintptr_t)workspace & 15) == 0);
  hipStream_t s = (hipStream_t)stream;
  const int c8 = dx->c >> 3;
  hipLaunchKernelGGL(upsample8_bwd_blocks_kernel, dim3((unsigned)(((dx->w + 1) * c8 + 255) / 256), (unsigned)(dx->h + 1), (unsigned)dx->n),
                     dim3(256), 0, s, (const __bf16*)dy->data, (float*)workspace, dx->n, dx->h, dx->w, dx->c);
  hipLaunchKernelGGL(upsample8_bwd_combine_kernel, dim3((unsigned)((dx->w * c8 + 255) / 256), (unsigned)dx->h, (unsigned)dx->n), dim3(256), 0,
                     s, (const float*)workspace, (__bf16*)dx->data, dx->n, dx->h, dx->w, dx->c);
  return xv_launch_status();
}

#define XV_CM_SWITCH(C_, CALL) \
  switch (((C_) + 3) / 4) {    \
    case 1: CALL(4); break;    \
    case 2: CALL(8); break;    \
    case 3: CALL(12); break;   \
    case 4: CALL(16); break;   \
    case 5: CALL(20); break;   \
    case 6: CALL(24); break;   \
    case 7: CALL(28); break;   \
    default: CALL(32); break;  \
  }

extern "C" int xv_score_dense_fwd(const xv_act* u, const float* w_score, const float* b_score, int num_classes,
                                  float* score, void* stream) {
  XV_REQUIRE_BF16(u);
  XV_CHECK_ARG(u && u->data && w_score && b_score && score);
  XV_CHECK_SHAPE((u->c & 7) == 0 && u->c <= 256 && num_classes >= 1 && num_classes <= 32);
  const int64_t npix = (int64_t)u->n * u->h * u->w;
  hipStream_t s = (hipStream_t)stream;
  // the matrix-core form for the FCN's head (64 channels, at most 16 classes); XV_SCORE_DENSE_OLD=1: the FMA kernels (A/B)
  static const bool sd_old = getenv("XV_SCORE_DENSE_OLD") != nullptr;
  if (u->c == 64 && num_classes <= 16 && npix < 0x7fff0000 && !sd_old) {
    hipLaunchKernelGGL(score_dense_mfma_kernel<2>, dim3(bn_grid(npix / 4 + 1, 2048)), dim3(256), 0, s, (const __bf16*)u->data,
                       w_score, b_score, score, u->n, u->h, u->w, num_classes);
    return xv_launch_status();
  }
#define XV_SD(CMV)                                                                                                   \
  hipLaunchKernelGGL(score_dense_kernel<CMV>, dim3(bn_grid(npix, 4096)), dim3(256), (size_t)u->c * CMV * 4, s,        \
                     (const __bf16*)u->data, w_score, b_score, score, u->n, u->h, u->w, u->c, num_classes)
  XV_CM_SWITCH(num_classes, XV_SD)
#undef XV_SD
  return xv_launch_status();
}

// y = relu(bilinear_x8(low) * scale + shift) (the forward apply pass of the batch norm behind the x8 deconv; what
// xv_bn_apply_ups8 writes, bit for bit) AND score = y . W + b (xv_score_dense_fwd's matrix-core form, bit for bit) in one
// launch: 64 units, at most 16 classes; XV_ESHAPE elsewhere (use the two calls).
extern "C" int xv_score_dense_fwd_ups8(const xv_act* low, const float* scale, const float* shift, const float* w_score,
                                       const float* b_score, int num_classes, const xv_act* y, float* score, void* stream) {
  XV_REQUIRE_BF16(low, y);
  XV_CHECK_ARG(low && low->data && y && y->data && scale && shift && w_score && b_score && score);
  const int64_t npix = (int64_t)y->n * y->h * y->w;
  if (y->c != 64 || num_classes < 1 || num_classes > 16 || npix >= 0x7fff0000) return XV_ESHAPE;
  XV_CHECK_SHAPE(low->n == y->n && 8 * low->h == y->h && 8 * low->w == y->w && low->c == 64);
  hipLaunchKernelGGL(score_dense_ups8_mfma_kernel, dim3(bn_grid(npix / 4 + 1, 2048)), dim3(256), 0, (hipStream_t)stream,
                     (const __bf16*)low->data, scale, shift, w_score, b_score, (__bf16*)y->data, score, y->n, y->h, y->w, num_classes);
  return xv_launch_status();
}

extern "C" size_t xv_softmax_ce_dense_workspace_bytes(int64_t npix) {
  return npix > 0 ? (size_t)bn_grid(npix, 2048) * sizeof(double) : 0;
}

extern "C" int xv_softmax_ce_dense_ws(const float* scores, const float* scale, const float* shift, const int32_t* labels,
                                      const int64_t* valid_count, int num_classes, int64_t npix, double* loss, float* dlogits,
                                      void* ws, size_t ws_bytes, void* stream) {
  XV_CHECK_ARG(scores && labels && valid_count && loss && dlogits && (scale == nullptr) == (shift == nullptr));
  XV_CHECK_SHAPE(npix > 0 && num_classes >= 1 && num_classes <= 32);
  XV_CHECK_ARG(ws == nullptr || (ws_bytes >= xv_softmax_ce_dense_workspace_bytes(npix) && ((uintptr_t)ws & 7) == 0));
  const unsigned grid = (unsigned)bn_grid(npix, 2048);
#define XV_CE(CMV)                                                                                                   \
  hipLaunchKernelGGL(softmax_ce_dense_kernel<CMV>, dim3(grid), dim3(256), 0, (hipStream_t)stream, scores, labels,     \
                     reinterpret_cast<const unsigned long long*>(valid_count), num_classes, npix, loss, dlogits, scale, \
                     shift, (double*)ws)
  XV_CM_SWITCH(num_classes, XV_CE)
#undef XV_CE
  if (ws != nullptr)
    hipLaunchKernelGGL(loss_partials_sum_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const double*)ws, (int)grid, loss);
  return xv_launch_status();
}

extern "C" int xv_softmax_ce_dense_affine(const float* scores, const float* scale, const float* shift, const int32_t* labels,
                                          const int64_t* valid_count, int num_classes, int64_t npix, double* loss,
                                          float* dlogits, void* stream) {
  return xv_softmax_ce_dense_ws(scores, scale, shift, labels, valid_count, num_classes, npix, loss, dlogits, nullptr, 0, stream);
}

extern "C" int xv_softmax_ce_dense(const float* logits, const int32_t* labels, const int64_t* valid_count,
                                   int num_classes, int64_t npix, double* loss, float* dlogits, void* stream) {
  return xv_softmax_ce_dense_ws(logits, nullptr, nullptr, labels, valid_count, num_classes, npix, loss, dlogits, nullptr, 0, stream);
}

// workspace of xv_score_dense_bwd_ws: one row of 33 x 64 partial sums per workgroup of the filter-gradient kernel + their totals
static unsigned score_dense_wgrad_grid(int64_t npix, int& qpw) {
  const int64_t nquads = (npix + 3) / 4;
  qpw = (int)((nquads + 2047) / 2048);
  qpw = (qpw + 3) / 4 * 4;
  return (unsigned)((nquads + (int64_t)4 * qpw - 1) / ((int64_t)4 * qpw));
}
// conv_wgrad.hip: the bf16-split form (64 units, C <= 16, maps that tile in 8x32); the grid it launches, < 0: does not apply
int xv_launch_score_wgrad_split(const float* ds, const void* y, int n, int h, int w, int c, float* part, int query_only,
                                hipStream_t stream);
static bool score_wgrad_split_ok() {
  static const bool on = getenv("XV_SCORE_WGRAD_SPLIT") == nullptr || atoi(getenv("XV_SCORE_WGRAD_SPLIT")) != 0;
  return on;
}
extern "C" size_t xv_score_dense_bwd_workspace_bytes(int n, int h, int w) {
  if (!xv_dims_sane(n, h, w) || (int64_t)n * h * w >= 0x7fff0000) return 0;
  int qpw;
  size_t rows = (size_t)score_dense_wgrad_grid((int64_t)n * h * w, qpw) * 33 * 64;
  const int gs = xv_launch_score_wgrad_split(nullptr, nullptr, n, h, w, 16, nullptr, 1, nullptr);
  if (gs > 0 && (size_t)gs * 17 * 64 > rows) rows = (size_t)gs * 17 * 64;
  return rows * sizeof(float) + 33 * 64 * sizeof(double);
}

extern "C" int xv_score_dense_bwd_ws(const xv_act* u, const float* dscore, const float* w_score, int num_classes, float* dw_score,
                                     float* db_score, const xv_act* du, void* workspace, size_t workspace_bytes, void* stream);
extern "C" int xv_score_dense_bwd(const xv_act* u, const float* dscore, const float* w_score, int num_classes,
                                  float* dw_score, float* db_score, const xv_act* du, void* stream) {
  return xv_score_dense_bwd_ws(u, dscore, w_score, num_classes, dw_score, db_score, du, nullptr, 0, stream);
}

// workspace (xv_score_dense_bwd_workspace_bytes; 64 units and at most 16 classes -- the matrix-core form): the filter and bias
// gradients are added in a fixed order instead of by atomics in arrival order: bitwise reproducible from run to run
extern "C" int xv_score_dense_bwd_ws(const xv_act* u, const float* dscore, const float* w_score, int num_classes, float* dw_score,
                                     float* db_score, const xv_act* du, void* workspace, size_t workspace_bytes, void* stream) {
  XV_REQUIRE_BF16(u, du);
  XV_CHECK_ARG(u && u->data && dscore && w_score && dw_score && db_score && du && du->data);
  XV_CHECK_SHAPE((u->c & 63) == 0 && u->c <= 256 && num_classes >= 1 && num_classes <= 32 && same_shape(u, du));
  const int64_t npix = (int64_t)u->n * u->h * u->w;
  // at most 512 workgroups: each ends with U*C same-address global atomics, which cross the XCDs' L2s and serialise
  // (1152 workgroups spent most of this kernel's 0.5 ms there)
  int64_t per_block = (npix + 511) / 512;
  per_block = (per_block + 255) / 256 * 256;
  if (per_block < 2048) per_block = 2048;
  const unsigned gw = (unsigned)((npix + per_block - 1) / per_block);
  XV_CHECK_SHAPE(npix < 0x7fff0000);
  const size_t lds = (size_t)SD_TILE * (u->c + 8) * 2;
  const unsigned gd = (unsigned)bn_grid(npix, 2048);
  const unsigned gm = (unsigned)bn_grid(npix / 2 + 1, 2048);
  static const bool sd_old = getenv("XV_SCORE_DENSE_OLD") != nullptr;
  const bool mfma_dgrad = u->c == 64 && num_classes <= 16 && !sd_old;  // exact fp32 on v_mfma_f32_16x16x4_f32
  static const bool wg_old = getenv("XV_SCORE_WGRAD_OLD") != nullptr;
  // filter gradient: at most 512 workgroups (each ends with ~800 same-address atomics); a wave walks a contiguous run
  int qpw;
  const unsigned gwm = score_dense_wgrad_grid(npix, qpw);
  hipStream_t s = (hipStream_t)stream;
  float* part = nullptr;
  if (workspace != nullptr && mfma_dgrad && !wg_old) {
    if (workspace_bytes < xv_score_dense_bwd_workspace_bytes(u->n, u->h, u->w)) return XV_EWORKSPACE;
    XV_CHECK_ARG(((uintptr_t)workspace & 15) == 0);
    part = (float*)workspace;
  }
#define XV_SB(CMV)                                                                                                   \
  {                                                                                                                  \
    static bool attr[XV_MAX_DEVICES] = {false};                                                                      \
    (void)xv_allow_dynamic_lds(reinterpret_cast<const void*>(&score_dense_dgrad_kernel<CMV>), 160 * 1024, attr, false);      \
    const int gs = (part != nullptr && score_wgrad_split_ok())                                                       \
                       ? xv_launch_score_wgrad_split(dscore, u->data, u->n, u->h, u->w, num_classes, part, 0, s) : -1;  \
    if (gs == -2) return XV_EINVAL;                                                                                  \
    if (gs > 0) {                                                                                                    \
      double* const tot = reinterpret_cast<double*>(part + (size_t)gs * 17 * 64);                                    \
      hipLaunchKernelGGL(bn_sums_kernel, dim3(17 * 64), dim3(256), 0, s, (const float*)part, gs, 17 * 64, tot,         \
                         (float*)nullptr, (float*)nullptr);                                                          \
      hipLaunchKernelGGL(score_dense_wgrad_split_scatter_kernel, dim3(1), dim3(256), 0, s, (const double*)tot,         \
                         num_classes, dw_score, db_score);                                                           \
    } else if (mfma_dgrad && !wg_old) {                                                                              \
      hipLaunchKernelGGL(score_dense_wgrad_mfma_kernel<8>, dim3(gwm), dim3(256), 0, s, (const __bf16*)u->data, dscore, \
                         dw_score, db_score, u->n, u->h, u->w, num_classes, qpw, part);                               \
      if (part != nullptr) {                                                                                         \
        double* const tot = reinterpret_cast<double*>(part + (size_t)gwm * 33 * 64);                                 \
        hipLaunchKernelGGL(bn_sums_kernel, dim3(33 * 64), dim3(256), 0, s, (const float*)part, (int)gwm, 33 * 64, tot, \
                           (float*)nullptr, (float*)nullptr);                                                        \
        hipLaunchKernelGGL(score_dense_wgrad_scatter_kernel, dim3(1), dim3(256), 0, s, (const double*)tot, num_classes, \
                           dw_score, db_score);                                                                      \
      }                                                                                                              \
    } else {                                                                                                         \
      hipLaunchKernelGGL(score_dense_wgrad_kernel<CMV>, dim3(gw), dim3(256), 0, s, (const __bf16*)u->data, dscore,    \
                         dw_score, db_score, u->n, u->h, u->w, u->c, num_classes, per_block);                         \
    }                                                                                                                \
    if (mfma_dgrad) {                                                                                                \
      if (num_classes <= 12)                                                                                         \
        hipLaunchKernelGGL((score_dense_dgrad_mfma_kernel<4, 3>), dim3(gm), dim3(256), 0, s, dscore, w_score,          \
                           (__bf16*)du->data, u->n, u->h, u->w, num_classes);                                         \
      else                                                                                                           \
        hipLaunchKernelGGL((score_dense_dgrad_mfma_kernel<4, 4>), dim3(gm), dim3(256), 0, s, dscore, w_score,          \
                           (__bf16*)du->data, u->n, u->h, u->w, num_classes);                                         \
    } else {                                                                                                         \
      hipLaunchKernelGGL(score_dense_dgrad_kernel<CMV>, dim3(gd), dim3(256), lds, s, dscore, w_score,                 \
                         (__bf16*)du->data, u->n, u->h, u->w, u->c, num_classes);                                     \
    }                                                                                                                \
  }
  XV_CM_SWITCH(num_classes, XV_SB)
#undef XV_SB
  return xv_launch_status();
}
