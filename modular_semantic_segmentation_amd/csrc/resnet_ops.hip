// Data-movement kernels that put AdapNet's strided and dilated convolutions onto the stride-1 MFMA conv kernels
// (adapnet.py:12-173).  All three are pure gathers over padded-NHWC bf16 maps, 16 bytes per thread, HBM-bound.
//
//  * xv_subsample2      y[i][j] = x[2i][2j]: the input of a 1x1 stride-2 conv ([TF1] 'same', no padding).
//  * xv_gather_conv7s2  the 7x7 stride-2 conv as ONE 3x3 stride-1 conv over 9*C channels.  With [TF1] 'same' padding
//                       (2 before) out(o) = sum_k W[k] x[2o + k - 2]; writing k = 2(t + s) + p, t in {0,1,2} the 3x3 tap,
//                       p the pixel parity and s in {0,1} an extra whole-pixel shift, the row variants
//                       (p,s) = (0,0) -> k = 0,2,4; (0,1) -> k = 6 (tap t = 2 only); (1,0) -> k = 1,3,5 cover k = 0..6
//                       exactly once.  Rows x columns = 9 channel groups; z[j][i][g] = x[2(j+sr)+pr][2(i+sc)+pc].
//                       The host scatters the 7x7 kernel into the [3][3][9C][Cout] kernel accordingly.
//  * xv_im2col_dilated_pair  block_b's two atrous 3x3 convs on the same input (rates d1, d2) as ONE 1x1 conv over
//                       18*C channels: z[y][x][t] = x[y + ty*d][x + tx*d], t = 0..8 at rate d1, 9..17 at rate d2, zero
//                       outside the image; the host stacks the two kernels block-wise so that the concatenated
//                       output of adapnet.py:84-88 comes out of one launch.
#include "xv_common.h"

namespace {

inline int rn_grid(int64_t total, int cap = 8192) {
  int64_t g = (total + 255) / 256;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

__global__ __launch_bounds__(256) void subsample2_kernel(const u32x4* __restrict__ x, u32x4* __restrict__ y, int N,
                                                        int Ho, int Wo, int c8) {
  const int Hi = Ho * 2, Wi = Wo * 2;
  const int64_t total = (int64_t)N * Ho * Wo * c8;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int cg = (int)(idx % c8);
    int64_t r = idx / c8;
    const int j = (int)(r % Wo);
    r /= Wo;
    const int i = (int)(r % Ho);
    const int n = (int)(r / Ho);
    y[(((int64_t)n * (Ho + 2) + i + 1) * (Wo + 2) + j + 1) * c8 + cg] =
        x[(((int64_t)n * (Hi + 2) + 2 * i + 1) * (Wi + 2) + 2 * j + 1) * c8 + cg];
  }
}

__global__ __launch_bounds__(256) void gather_conv7s2_kernel(const u32x4* __restrict__ x, u32x4* __restrict__ z, int N,
                                                            int Ho, int Wo, int c8) {
  const int Hi = Ho * 2, Wi = Wo * 2;
  const int g8 = 9 * c8;
  const int64_t total = (int64_t)N * Ho * Wo * g8;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int gc = (int)(idx % g8);
    int64_t r = idx / g8;
    const int i = (int)(r % Wo);
    r /= Wo;
    const int j = (int)(r % Ho);
    const int n = (int)(r / Ho);
    const int g = gc / c8, cg = gc - g * c8;
    const int rv = g / 3, cv = g - rv * 3;  // variants 0: (p=0,s=0), 1: (p=0,s=1), 2: (p=1,s=0)
    const int sy = 2 * j + (rv == 1 ? 2 : (rv == 2 ? 1 : 0));
    const int sx = 2 * i + (cv == 1 ? 2 : (cv == 2 ? 1 : 0));
    u32x4 v = u32x4{0u, 0u, 0u, 0u};
    if (sy < Hi && sx < Wi) v = x[(((int64_t)n * (Hi + 2) + sy + 1) * (Wi + 2) + sx + 1) * c8 + cg];
    z[(((int64_t)n * (Ho + 2) + j + 1) * (Wo + 2) + i + 1) * g8 + gc] = v;
  }
}

__global__ __launch_bounds__(256) void im2col_pair_kernel(const u32x4* __restrict__ x, u32x4* __restrict__ z, int N,
                                                         int H, int W, int c8, int d1, int d2) {
  const int g8 = 18 * c8;
  const int64_t total = (int64_t)N * H * W * g8;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int gc = (int)(idx % g8);
    int64_t r = idx / g8;
    const int px = (int)(r % W);
    r /= W;
    const int py = (int)(r % H);
    const int n = (int)(r / H);
    const int t = gc / c8, cg = gc - t * c8;
    const int d = t < 9 ? d1 : d2;
    const int tt = t < 9 ? t : t - 9;
    const int sy = py + (tt / 3 - 1) * d, sx = px + (tt % 3 - 1) * d;
    u32x4 v = u32x4{0u, 0u, 0u, 0u};
    if (sy >= 0 && sy < H && sx >= 0 && sx < W) v = x[(((int64_t)n * (H + 2) + sy + 1) * (W + 2) + sx + 1) * c8 + cg];
    z[(((int64_t)n * (H + 2) + py + 1) * (W + 2) + px + 1) * g8 + gc] = v;
  }
}

// Row forms of the two gathers above (the defaults): a destination row (n, y) is CONTIGUOUS in z (x-major, then tap, then
// channel group), so a workgroup walks 1024 consecutive 16-byte elements of one row and only has to split the in-row index
// into (x, tap, channel group).  The flat forms do that with three 64-bit divisions per 16 bytes -- a few hundred
// instructions per element, which made a pure copy ALU-bound at 2.7 TB/s of writes; here the two divisions are one
// v_mul_hi each by a reciprocal the launcher computed (exact while row_length * divisor < 2^32, which it checks).
__device__ __forceinline__ uint32_t div_magic(uint32_t n, uint32_t magic) { return __umulhi(n, magic); }

__global__ __launch_bounds__(256) void im2col_pair_rows_kernel(const u32x4* __restrict__ x, u32x4* __restrict__ z, int H,
                                                              int W, int c8, uint32_t magic_g8, uint32_t magic_c8, int d1,
                                                              int d2) {
  const uint32_t g8 = 18u * (uint32_t)c8, L = (uint32_t)W * g8;
  const int row = blockIdx.y, n = row / H, py = row - n * H;
  const u32x4* xn = x + (int64_t)n * (H + 2) * (W + 2) * c8;
  u32x4* zr = z + (((int64_t)n * (H + 2) + py + 1) * (W + 2) + 1) * g8;
  const uint32_t base = blockIdx.x * 1024u + threadIdx.x;
  u32x4 v[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const uint32_t i = base + 256u * u;
    v[u] = u32x4{0u, 0u, 0u, 0u};
    if (i < L) {
      const uint32_t px = div_magic(i, magic_g8), gc = i - px * g8;
      const uint32_t t = div_magic(gc, magic_c8), cg = gc - t * (uint32_t)c8;
      const int d = t < 9u ? d1 : d2;
      const int tt = t < 9u ? (int)t : (int)t - 9;
      const int ky = tt / 3, kx = tt - 3 * ky;
      const int sy = py + (ky - 1) * d, sx = (int)px + (kx - 1) * d;
      if (sy >= 0 && sy < H && sx >= 0 && sx < W) v[u] = xn[((int64_t)(sy + 1) * (W + 2) + sx + 1) * c8 + cg];
    }
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const uint32_t i = base + 256u * u;
    if (i < L) zr[i] = v[u];
  }
}

__global__ __launch_bounds__(256) void gather_conv7s2_rows_kernel(const u32x4* __restrict__ x, u32x4* __restrict__ z,
                                                                 int Ho, int Wo, int c8, uint32_t magic_g8,
                                                                 uint32_t magic_c8) {
  const int Hi = Ho * 2, Wi = Wo * 2;
  const uint32_t g8 = 9u * (uint32_t)c8, L = (uint32_t)Wo * g8;
  const int row = blockIdx.y, n = row / Ho, j = row - n * Ho;
  const u32x4* xn = x + (int64_t)n * (Hi + 2) * (Wi + 2) * c8;
  u32x4* zr = z + (((int64_t)n * (Ho + 2) + j + 1) * (Wo + 2) + 1) * g8;
  const uint32_t base = blockIdx.x * 1024u + threadIdx.x;
  u32x4 v[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const uint32_t i = base + 256u * u;
    v[u] = u32x4{0u, 0u, 0u, 0u};
    if (i < L) {
      const uint32_t px = div_magic(i, magic_g8), gc = i - px * g8;
      const uint32_t g = div_magic(gc, magic_c8), cg = gc - g * (uint32_t)c8;
      const int rv = (int)g / 3, cv = (int)g - rv * 3;
      const int sy = 2 * j + (rv == 1 ? 2 : (rv == 2 ? 1 : 0));
      const int sx = 2 * (int)px + (cv == 1 ? 2 : (cv == 2 ? 1 : 0));
      if (sy < Hi && sx < Wi) v[u] = xn[((int64_t)(sy + 1) * (Wi + 2) + sx + 1) * c8 + cg];
    }
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const uint32_t i = base + 256u * u;
    if (i < L) zr[i] = v[u];
  }
}

// floor(2^32 / d) + 1: __umulhi(n, magic) == n / d for every n with n * d < 2^32
inline uint32_t magic_of(uint32_t d) { return (uint32_t)((((uint64_t)1) << 32) / d) + 1u; }
inline bool rows_form_ok(int64_t rows, int64_t row_len, int64_t g8) {
  static const bool flat = getenv("XV_GATHER_FLAT") != nullptr && atoi(getenv("XV_GATHER_FLAT")) != 0;  // A/B switch
  return rows <= 65535 && row_len * g8 < (((int64_t)1) << 32) && !flat;
}

// ---- transposes of the three gathers (training) and the residual add, all written as gathers themselves: every
// destination element is produced by exactly one thread, no atomics, deterministic -------------------------------------
__device__ __forceinline__ void acc_bf16x8(float (&s)[8], const u32x4 v) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    s[2 * i] += bf16_bits_to_f32(v[i] & 0xffffu);
    s[2 * i + 1] += __builtin_bit_cast(float, v[i] & 0xffff0000u);
  }
}
__device__ __forceinline__ u32x4 pack8(const float (&s)[8]) {
  return u32x4{pack_bf16x2(s[0], s[1]), pack_bf16x2(s[2], s[3]), pack_bf16x2(s[4], s[5]), pack_bf16x2(s[6], s[7])};
}

// dx[2i][2j] = dy[i][j], zero elsewhere
__global__ __launch_bounds__(256) void subsample2_bwd_kernel(const u32x4* __restrict__ dy, u32x4* __restrict__ dx, int N,
                                                            int Ho, int Wo, int c8) {
  const int Hi = Ho * 2, Wi = Wo * 2;
  const int64_t total = (int64_t)N * Hi * Wi * c8;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int cg = (int)(idx % c8);
    int64_t r = idx / c8;
    const int x = (int)(r % Wi);
    r /= Wi;
    const int y = (int)(r % Hi);
    const int n = (int)(r / Hi);
    u32x4 v = u32x4{0u, 0u, 0u, 0u};
    if (((x | y) & 1) == 0) v = dy[(((int64_t)n * (Ho + 2) + (y >> 1) + 1) * (Wo + 2) + (x >> 1) + 1) * c8 + cg];
    dx[(((int64_t)n * (Hi + 2) + y + 1) * (Wi + 2) + x + 1) * c8 + cg] = v;
  }
}

// dx[sy][sx] = sum over the (row variant, column variant) pairs that read this pixel of dz[j][i][group]:
// an even coordinate s is read by variant 0 at j = s/2 and by variant 1 (shift) at j = s/2 - 1, an odd one by variant 2
__global__ __launch_bounds__(256) void gather_conv7s2_bwd_kernel(const u32x4* __restrict__ dz, u32x4* __restrict__ dx,
                                                                int N, int Ho, int Wo, int c8) {
  const int Hi = Ho * 2, Wi = Wo * 2;
  const int g8 = 9 * c8;
  const int64_t total = (int64_t)N * Hi * Wi * c8;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int cg = (int)(idx % c8);
    int64_t r = idx / c8;
    const int sx = (int)(r % Wi);
    r /= Wi;
    const int sy = (int)(r % Hi);
    const int n = (int)(r / Hi);
    int rvs[2], js[2], nr = 0, cvs[2], is[2], nc = 0;
    if (sy & 1) {
      rvs[nr] = 2, js[nr++] = sy >> 1;
    } else {
      rvs[nr] = 0, js[nr++] = sy >> 1;
      if (sy >= 2) rvs[nr] = 1, js[nr++] = (sy >> 1) - 1;
    }
    if (sx & 1) {
      cvs[nc] = 2, is[nc++] = sx >> 1;
    } else {
      cvs[nc] = 0, is[nc++] = sx >> 1;
      if (sx >= 2) cvs[nc] = 1, is[nc++] = (sx >> 1) - 1;
    }
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int a = 0; a < nr; ++a)
      for (int b = 0; b < nc; ++b)
        acc_bf16x8(s, dz[(((int64_t)n * (Ho + 2) + js[a] + 1) * (Wo + 2) + is[b] + 1) * g8 + (rvs[a] * 3 + cvs[b]) * c8 + cg]);
    dx[(((int64_t)n * (Hi + 2) + sy + 1) * (Wi + 2) + sx + 1) * c8 + cg] = pack8(s);
  }
}

// col2im: dx[y][x] = sum_t dz[y - ty*d][x - tx*d][t] over the 18 taps whose reader lies inside the image
__global__ __launch_bounds__(256) void im2col_pair_bwd_kernel(const u32x4* __restrict__ dz, u32x4* __restrict__ dx, int N,
                                                             int H, int W, int c8, int d1, int d2) {
  const int g8 = 18 * c8;
  const int64_t total = (int64_t)N * H * W * c8;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int cg = (int)(idx % c8);
    int64_t r = idx / c8;
    const int px = (int)(r % W);
    r /= W;
    const int py = (int)(r % H);
    const int n = (int)(r / H);
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < 18; ++t) {
      const int d = t < 9 ? d1 : d2;
      const int tt = t < 9 ? t : t - 9;
      const int ry = py - (tt / 3 - 1) * d, rx = px - (tt % 3 - 1) * d;
      if (ry >= 0 && ry < H && rx >= 0 && rx < W)
        acc_bf16x8(s, dz[(((int64_t)n * (H + 2) + ry + 1) * (W + 2) + rx + 1) * g8 + t * c8 + cg]);
    }
    dx[(((int64_t)n * (H + 2) + py + 1) * (W + 2) + px + 1) * c8 + cg] = pack8(s);
  }
}

__global__ __launch_bounds__(256) void add_kernel(const u32x4* __restrict__ a, const u32x4* __restrict__ b,
                                                 u32x4* __restrict__ y, int64_t total) {
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    acc_bf16x8(s, a[idx]);
    acc_bf16x8(s, b[idx]);
    y[idx] = pack8(s);
  }
}

// ---- phase shuffles of the DENSE transposed convs (AdapNet trains its two deconv kernels: adapnet.py:155-163 calls
// custom_layers.deconv2d:71-121 without trainable=False).  A k = 2s transposed conv is one 3x3 conv onto s*s*C "phase"
// channels at the input resolution (custom_layers.dense_deconv_as_conv3x3) followed by a depth-to-space shuffle; its
// gradients are the 3x3 conv's filter / data gradients of the space-to-depth shuffle of the upstream gradient.
// Phase channel layout: (py*s + px)*C + c  <->  output pixel (s*qy + py, s*qx + px), channel c.

// g [N][s*Hq][s*Wq][C] (padded NHWC bf16) -> out [N][Hq][Wq][s*s*C]: 16 bytes per thread
__global__ __launch_bounds__(256) void space_to_depth_kernel(const u32x4* __restrict__ g, u32x4* __restrict__ out, int N,
                                                            int Hq, int Wq, int c8, int S) {
  const int Ho = Hq * S, Wo = Wq * S;
  const int pc8 = S * S * c8;
  const int64_t total = (int64_t)N * Hq * Wq * pc8;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int pc = (int)(idx % pc8);
    int64_t r = idx / pc8;
    const int qx = (int)(r % Wq);
    r /= Wq;
    const int qy = (int)(r % Hq);
    const int n = (int)(r / Hq);
    const int ph = pc / c8, cg = pc - ph * c8;
    const int py = ph / S, px = ph - py * S;
    out[(((int64_t)n * (Hq + 2) + qy + 1) * (Wq + 2) + qx + 1) * pc8 + pc] =
        g[(((int64_t)n * (Ho + 2) + qy * S + py + 1) * (Wo + 2) + qx * S + px + 1) * c8 + cg];
  }
}

// the same from a dense float32 [N][s*Hq][s*Wq][C] gradient (the class scores): channels C .. Cp-1 of every phase are zero
__global__ __launch_bounds__(256) void space_to_depth_dense_kernel(const float* __restrict__ g, u32x4* __restrict__ out,
                                                                  int N, int Hq, int Wq, int C, int c8, int S) {
  const int Ho = Hq * S, Wo = Wq * S;
  const int pc8 = S * S * c8;
  const int64_t total = (int64_t)N * Hq * Wq * pc8;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int pc = (int)(idx % pc8);
    int64_t r = idx / pc8;
    const int qx = (int)(r % Wq);
    r /= Wq;
    const int qy = (int)(r % Hq);
    const int n = (int)(r / Hq);
    const int ph = pc / c8, cg = pc - ph * c8;
    const int py = ph / S, px = ph - py * S;
    const float* src = g + (((int64_t)n * Ho + qy * S + py) * Wo + qx * S + px) * C;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = cg * 8 + j < C ? src[cg * 8 + j] : 0.f;
    out[(((int64_t)n * (Hq + 2) + qy + 1) * (Wq + 2) + qx + 1) * pc8 + pc] =
        u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
  }
}

// phase map z [N][Hq][Wq][s*s*Cp] -> dense float32 [N][s*Hq][s*Wq][C] (the class scores), optional per-class affine
// (inference batch norm of the x8 deconv): one thread per output pixel
__global__ __launch_bounds__(256) void depth_to_space_dense_kernel(const __bf16* __restrict__ z,
                                                                  const float* __restrict__ scale,
                                                                  const float* __restrict__ shift, float* __restrict__ out,
                                                                  int N, int Hq, int Wq, int C, int Cp, int S) {
  const int Ho = Hq * S, Wo = Wq * S;
  const int64_t total = (int64_t)N * Ho * Wo;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int ox = (int)(idx % Wo);
    int64_t r = idx / Wo;
    const int oy = (int)(r % Ho);
    const int n = (int)(r / Ho);
    const int qy = oy / S, py = oy - qy * S, qx = ox / S, px = ox - qx * S;
    const __bf16* src = z + (((int64_t)n * (Hq + 2) + qy + 1) * (Wq + 2) + qx + 1) * ((int64_t)S * S * Cp) +
                        (int64_t)(py * S + px) * Cp;
    float* dst = out + idx * C;
    for (int c0 = 0; c0 < C; c0 += 8) {
      const u32x4 a = *reinterpret_cast<const u32x4*>(src + c0);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (c0 + j < C) {
          float v = (j & 1) ? __builtin_bit_cast(float, a[j >> 1] & 0xffff0000u) : bf16_bits_to_f32(a[j >> 1] & 0xffffu);
          if (scale != nullptr) v = v * scale[c0 + j] + shift[c0 + j];
          dst[c0 + j] = v;
        }
      }
    }
  }
}

// The same from a DENSE float32 phase map [N][Hq][Wq][s*s*Cp] (the x8 deconv of the class scores run in float32 on the fp32
// matrix instruction, adapnet.py:155-163: the reference computes the scores in float32, a bf16 phase map in between would
// round them to 8 bits before the batch norm and the softmax)
__global__ __launch_bounds__(256) void depth_to_space_dense_f32_kernel(const float* __restrict__ z, const float* __restrict__ scale,
                                                                      const float* __restrict__ shift, float* __restrict__ out,
                                                                      int N, int Hq, int Wq, int C, int Cp, int S) {
  const int Ho = Hq * S, Wo = Wq * S;
  const int64_t total = (int64_t)N * Ho * Wo;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int ox = (int)(idx % Wo);
    int64_t r = idx / Wo;
    const int oy = (int)(r % Ho);
    const int n = (int)(r / Ho);
    const int qy = oy / S, py = oy - qy * S, qx = ox / S, px = ox - qx * S;
    const float* src = z + (((int64_t)n * Hq + qy) * Wq + qx) * ((int64_t)S * S * Cp) + (int64_t)(py * S + px) * Cp;
    float* dst = out + idx * C;
    for (int c0 = 0; c0 < C; c0 += 4) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(src + c0);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (c0 + j < C) dst[c0 + j] = scale != nullptr ? a[j] * scale[c0 + j] + shift[c0 + j] : a[j];
    }
  }
}

// padded bf16 map -> dense float32 [N][H][W][C] (exact: every bf16 is a float), 8 channels per thread
__global__ __launch_bounds__(256) void act_to_dense_f32_kernel(const __bf16* __restrict__ x, float* __restrict__ out, int N, int H,
                                                              int W, int C8) {
  const int64_t total = (int64_t)N * H * W * C8;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(idx % C8);
    int64_t r = idx / C8;
    const int px = (int)(r % W);
    r /= W;
    const int py = (int)(r % H);
    const int n = (int)(r / H);
    const u32x4 a = *reinterpret_cast<const u32x4*>(x + ((((int64_t)n * (H + 2) + py + 1) * (W + 2) + px + 1) * C8 + c8) * 8);
    float* dst = out + idx * 8;
    *reinterpret_cast<f32x4*>(dst) = f32x4{bf16_bits_to_f32(a.x & 0xffffu), __builtin_bit_cast(float, a.x & 0xffff0000u),
                                           bf16_bits_to_f32(a.y & 0xffffu), __builtin_bit_cast(float, a.y & 0xffff0000u)};
    *reinterpret_cast<f32x4*>(dst + 4) = f32x4{bf16_bits_to_f32(a.z & 0xffffu), __builtin_bit_cast(float, a.z & 0xffff0000u),
                                               bf16_bits_to_f32(a.w & 0xffffu), __builtin_bit_cast(float, a.w & 0xffff0000u)};
  }
}

}  // namespace

extern "C" int xv_act_to_dense_f32(const xv_act* x, float* out, void* stream) {
  XV_REQUIRE_BF16(x);
  XV_CHECK_ARG(x && x->data && out && ((uintptr_t)out & 15) == 0);
  XV_CHECK_SHAPE(xv_dims_sane(x->n, x->h, x->w) && x->c > 0 && (x->c & 7) == 0);
  const int64_t total = (int64_t)x->n * x->h * x->w * (x->c >> 3);
  hipLaunchKernelGGL(act_to_dense_f32_kernel, dim3(rn_grid(total)), dim3(256), 0, (hipStream_t)stream, (const __bf16*)x->data, out,
                     x->n, x->h, x->w, x->c >> 3);
  return xv_launch_status();
}

extern "C" int xv_depth_to_space_dense_f32(const float* z, int n, int hq, int wq, int stride, int cp, int num_classes,
                                           const float* scale, const float* shift, float* out, void* stream) {
  XV_CHECK_ARG(z && out && (scale == nullptr) == (shift == nullptr) && ((uintptr_t)z & 15) == 0);
  XV_CHECK_SHAPE(n > 0 && hq > 0 && wq > 0 && stride >= 1 && stride <= 16 && num_classes >= 1 && (cp & 3) == 0 && cp >= num_classes);
  const int64_t total = (int64_t)n * hq * wq * stride * stride;
  hipLaunchKernelGGL(depth_to_space_dense_f32_kernel, dim3(rn_grid(total)), dim3(256), 0, (hipStream_t)stream, z, scale, shift, out,
                     n, hq, wq, num_classes, cp, stride);
  return xv_launch_status();
}

extern "C" int xv_subsample2(const xv_act* x, const xv_act* y, void* stream) {
  XV_REQUIRE_BF16(x, y);
  XV_CHECK_ARG(x && y && x->data && y->data);
  XV_CHECK_SHAPE(x->n == y->n && x->c == y->c && (x->c & 7) == 0 && x->h == 2 * y->h && x->w == 2 * y->w && y->h > 0);
  const int64_t total = (int64_t)y->n * y->h * y->w * (y->c >> 3);
  hipLaunchKernelGGL(subsample2_kernel, dim3(rn_grid(total)), dim3(256), 0, (hipStream_t)stream, (const u32x4*)x->data,
                     (u32x4*)y->data, y->n, y->h, y->w, y->c >> 3);
  return xv_launch_status();
}

extern "C" int xv_gather_conv7s2(const xv_act* x, const xv_act* z, void* stream) {
  XV_REQUIRE_BF16(x, z);
  XV_CHECK_ARG(x && z && x->data && z->data);
  XV_CHECK_SHAPE(x->n == z->n && z->c == 9 * x->c && (x->c & 7) == 0 && x->h == 2 * z->h && x->w == 2 * z->w &&
                 z->h > 0);
  const int64_t total = (int64_t)z->n * z->h * z->w * (z->c >> 3);
  const int c8 = x->c >> 3, g8 = 9 * c8;
  if (rows_form_ok((int64_t)z->n * z->h, (int64_t)z->w * g8, g8)) {
    hipLaunchKernelGGL(gather_conv7s2_rows_kernel, dim3((z->w * g8 + 1023) / 1024, z->n * z->h), dim3(256), 0,
                       (hipStream_t)stream, (const u32x4*)x->data, (u32x4*)z->data, z->h, z->w, c8, magic_of(g8),
                       magic_of(c8));
    return xv_launch_status();
  }
  hipLaunchKernelGGL(gather_conv7s2_kernel, dim3(rn_grid(total)), dim3(256), 0, (hipStream_t)stream,
                     (const u32x4*)x->data, (u32x4*)z->data, z->n, z->h, z->w, x->c >> 3);
  return xv_launch_status();
}

extern "C" int xv_im2col_dilated_pair(const xv_act* x, int dilation1, int dilation2, const xv_act* z, void* stream) {
  XV_REQUIRE_BF16(x, z);
  XV_CHECK_ARG(x && z && x->data && z->data);
  XV_CHECK_SHAPE(x->n == z->n && x->h == z->h && x->w == z->w && z->c == 18 * x->c && (x->c & 7) == 0 &&
                 dilation1 >= 1 && dilation2 >= 1);
  const int64_t total = (int64_t)z->n * z->h * z->w * (z->c >> 3);
  const int c8 = x->c >> 3, g8 = 18 * c8;
  if (rows_form_ok((int64_t)z->n * z->h, (int64_t)z->w * g8, g8)) {
    hipLaunchKernelGGL(im2col_pair_rows_kernel, dim3((z->w * g8 + 1023) / 1024, z->n * z->h), dim3(256), 0,
                       (hipStream_t)stream, (const u32x4*)x->data, (u32x4*)z->data, z->h, z->w, c8, magic_of(g8),
                       magic_of(c8), dilation1, dilation2);
    return xv_launch_status();
  }
  hipLaunchKernelGGL(im2col_pair_kernel, dim3(rn_grid(total)), dim3(256), 0, (hipStream_t)stream, (const u32x4*)x->data,
                     (u32x4*)z->data, z->n, z->h, z->w, x->c >> 3, dilation1, dilation2);
  return xv_launch_status();
}

extern "C" int xv_subsample2_bwd(const xv_act* dy, const xv_act* dx, void* stream) {
  XV_REQUIRE_BF16(dy, dx);
  XV_CHECK_ARG(dy && dx && dy->data && dx->data);
  XV_CHECK_SHAPE(dx->n == dy->n && dx->c == dy->c && (dx->c & 7) == 0 && dx->h == 2 * dy->h && dx->w == 2 * dy->w &&
                 dy->h > 0);
  const int64_t total = (int64_t)dx->n * dx->h * dx->w * (dx->c >> 3);
  hipLaunchKernelGGL(subsample2_bwd_kernel, dim3(rn_grid(total)), dim3(256), 0, (hipStream_t)stream,
                     (const u32x4*)dy->data, (u32x4*)dx->data, dy->n, dy->h, dy->w, dy->c >> 3);
  return xv_launch_status();
}

extern "C" int xv_gather_conv7s2_bwd(const xv_act* dz, const xv_act* dx, void* stream) {
  XV_REQUIRE_BF16(dz, dx);
  XV_CHECK_ARG(dz && dx && dz->data && dx->data);
  XV_CHECK_SHAPE(dx->n == dz->n && dz->c == 9 * dx->c && (dx->c & 7) == 0 && dx->h == 2 * dz->h && dx->w == 2 * dz->w &&
                 dz->h > 0);
  const int64_t total = (int64_t)dx->n * dx->h * dx->w * (dx->c >> 3);
  hipLaunchKernelGGL(gather_conv7s2_bwd_kernel, dim3(rn_grid(total)), dim3(256), 0, (hipStream_t)stream,
                     (const u32x4*)dz->data, (u32x4*)dx->data, dz->n, dz->h, dz->w, dx->c >> 3);
  return xv_launch_status();
}

extern "C" int xv_im2col_dilated_pair_bwd(const xv_act* dz, int dilation1, int dilation2, const xv_act* dx,
                                          void* stream) {
  XV_REQUIRE_BF16(dz, dx);
  XV_CHECK_ARG(dz && dx && dz->data && dx->data);
  XV_CHECK_SHAPE(dx->n == dz->n && dx->h == dz->h && dx->w == dz->w && dz->c == 18 * dx->c && (dx->c & 7) == 0 &&
                 dilation1 >= 1 && dilation2 >= 1);
  const int64_t total = (int64_t)dx->n * dx->h * dx->w * (dx->c >> 3);
  hipLaunchKernelGGL(im2col_pair_bwd_kernel, dim3(rn_grid(total)), dim3(256), 0, (hipStream_t)stream,
                     (const u32x4*)dz->data, (u32x4*)dx->data, dx->n, dx->h, dx->w, dx->c >> 3, dilation1, dilation2);
  return xv_launch_status();
}

// y = a + b over the whole padded buffers (the residual add of a ResNet block in the training graph, where a batch
// norm sits between the conv and the add; borders are 0 + 0)
extern "C" int xv_add(const xv_act* a, const xv_act* b, const xv_act* y, void* stream) {
  XV_REQUIRE_BF16(a, b, y);
  XV_CHECK_ARG(a && b && y && a->data && b->data && y->data);
  XV_CHECK_SHAPE(a->n == b->n && a->h == b->h && a->w == b->w && a->c == b->c && y->n == a->n && y->h == a->h &&
                 y->w == a->w && y->c == a->c && (a->c & 7) == 0);
  const int64_t total = (int64_t)a->n * (a->h + 2) * (a->w + 2) * (a->c >> 3);
  hipLaunchKernelGGL(add_kernel, dim3(rn_grid(total)), dim3(256), 0, (hipStream_t)stream, (const u32x4*)a->data,
                     (const u32x4*)b->data, (u32x4*)y->data, total);
  return xv_launch_status();
}

extern "C" int xv_space_to_depth(const xv_act* g, int stride, const xv_act* out, void* stream) {
  XV_REQUIRE_BF16(g, out);
  XV_CHECK_ARG(g && out && g->data && out->data);
  XV_CHECK_SHAPE(stride >= 1 && stride <= 16 && g->n == out->n && g->h == stride * out->h && g->w == stride * out->w &&
                 out->c == stride * stride * g->c && (g->c & 7) == 0 && out->h > 0);
  const int64_t total = (int64_t)out->n * out->h * out->w * (out->c >> 3);
  hipLaunchKernelGGL(space_to_depth_kernel, dim3(rn_grid(total)), dim3(256), 0, (hipStream_t)stream,
                     (const u32x4*)g->data, (u32x4*)out->data, out->n, out->h, out->w, g->c >> 3, stride);
  return xv_launch_status();
}

extern "C" int xv_space_to_depth_dense(const float* g, int num_classes, int stride, const xv_act* out, void* stream) {
  XV_REQUIRE_BF16(out);
  XV_CHECK_ARG(g && out && out->data);
  XV_CHECK_SHAPE(stride >= 1 && stride <= 16 && num_classes >= 1 && out->c % (stride * stride) == 0 && out->h > 0);
  const int cp = out->c / (stride * stride);
  XV_CHECK_SHAPE((cp & 7) == 0 && cp >= num_classes);
  const int64_t total = (int64_t)out->n * out->h * out->w * (out->c >> 3);
  hipLaunchKernelGGL(space_to_depth_dense_kernel, dim3(rn_grid(total)), dim3(256), 0, (hipStream_t)stream, g,
                     (u32x4*)out->data, out->n, out->h, out->w, num_classes, cp >> 3, stride);
  return xv_launch_status();
}

extern "C" int xv_depth_to_space_dense(const xv_act* z, int stride, int num_classes, const float* scale, const float* shift,
                                       float* out, void* stream) {
  XV_REQUIRE_BF16(z);
  XV_CHECK_ARG(z && z->data && out && (scale == nullptr) == (shift == nullptr));
  XV_CHECK_SHAPE(stride >= 1 && stride <= 16 && num_classes >= 1 && z->c % (stride * stride) == 0 && z->h > 0);
  const int cp = z->c / (stride * stride);
  XV_CHECK_SHAPE((cp & 7) == 0 && cp >= num_classes);
  const int64_t total = (int64_t)z->n * z->h * z->w * stride * stride;
  hipLaunchKernelGGL(depth_to_space_dense_kernel, dim3(rn_grid(total)), dim3(256), 0, (hipStream_t)stream,
                     (const __bf16*)z->data, scale, shift, out, z->n, z->h, z->w, num_classes, cp, stride);
  return xv_launch_status();
}
