// HBM-bound layers of the FCN for gfx950: first conv (fp32 in), max-pool, bilinear x2 (+add),
// and the fused decoder head (bilinear x8 + relu + 1x1 score + softmax + argmax).
#include "xv_common.h"

namespace {

// ---- conv1_1: relu(conv3x3(x) + b) on the raw fp32 input, fp32 math, bf16 padded-NHWC out -------
// simple_fcn.py:39.  One thread = one pixel x 64 output channels; the 9*CIN x 64 fp32 weight
// matrix sits in LDS and is read with wave-uniform (broadcast) ds_read_b128.
template <int CIN>
__global__ __launch_bounds__(256) void conv_first_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ b, __bf16* __restrict__ y, int N,
                                                        int H, int W, int relu) {
  constexpr int K = 9 * CIN;
  __shared__ __attribute__((aligned(16))) float ws[K * 64];
  __shared__ __attribute__((aligned(16))) float bs[64];
  for (int i = threadIdx.x; i < K * 64; i += 256) ws[i] = w[i];
  if (threadIdx.x < 64) bs[threadIdx.x] = b[threadIdx.x];
  __syncthreads();
  const int64_t npix = (int64_t)N * H * W;
  const int64_t pix = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (pix >= npix) return;
  const int px = (int)(pix % W);
  const int py = (int)((pix / W) % H);
  const int n = (int)(pix / ((int64_t)W * H));
  float in[K];
#pragma unroll
  for (int dy = 0; dy < 3; ++dy)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int yy = py + dy - 1, xx = px + dx - 1;
      const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
      const float* src = x + (((int64_t)n * H + (ok ? yy : 0)) * W + (ok ? xx : 0)) * CIN;
#pragma unroll
      for (int c = 0; c < CIN; ++c) in[(dy * 3 + dx) * CIN + c] = ok ? src[c] : 0.f;
    }
  __bf16* dst = y + (((int64_t)n * (H + 2) + (py + 1)) * (W + 2) + (px + 1)) * 64;
#pragma unroll 1
  for (int g = 0; g < 8; ++g) {
    f32x4 a0 = *reinterpret_cast<const f32x4*>(bs + g * 8);
    f32x4 a1 = *reinterpret_cast<const f32x4*>(bs + g * 8 + 4);
#pragma unroll
    for (int t = 0; t < K; ++t) {
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(ws + t * 64 + g * 8);
      const f32x4 w1 = *reinterpret_cast<const f32x4*>(ws + t * 64 + g * 8 + 4);
      a0 += in[t] * w0;
      a1 += in[t] * w1;
    }
    if (relu) {
      a0.x = fmaxf(a0.x, 0.f); a0.y = fmaxf(a0.y, 0.f); a0.z = fmaxf(a0.z, 0.f); a0.w = fmaxf(a0.w, 0.f);
      a1.x = fmaxf(a1.x, 0.f); a1.y = fmaxf(a1.y, 0.f); a1.z = fmaxf(a1.z, 0.f); a1.w = fmaxf(a1.w, 0.f);
    }
    *reinterpret_cast<u32x4*>(dst + g * 8) =
        u32x4{pack_bf16x2(a0.x, a0.y), pack_bf16x2(a0.z, a0.w), pack_bf16x2(a1.x, a1.y), pack_bf16x2(a1.z, a1.w)};
  }
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
__global__ __launch_bounds__(256) void upsample2x_kernel(const __bf16* __restrict__ x, const __bf16* __restrict__ res,
                                                        __bf16* __restrict__ y, int N, int Hi, int Wi, int C) {
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
        u = fmaxf(u, 0.f);
        v[h] = u + bf16_bits_to_f32((rv[i] >> sh) & 0xffffu);
      }
      out[i] = pack_bf16x2(v[0], v[1]);
    }
    *reinterpret_cast<u32x4*>(y + oidx) = out;
  }
}

// ---- decoder head: bilinear x8 + relu + 1x1 score + softmax + argmax, one thread per pixel ------
// simple_fcn.py:129-133 + basic_fusion_model.py:21-22.  CMAX = padded class count (static register
// indexing); Ws is staged in LDS as [U][CMAX] and read with wave-uniform addresses.
template <int CMAX>
__global__ __launch_bounds__(256) void decoder_head_kernel(const __bf16* __restrict__ f, const float* __restrict__ ws_g,
                                                          const float* __restrict__ bs_g, int N, int Hi, int Wi, int U,
                                                          int C, float* __restrict__ score, float* __restrict__ prob,
                                                          int64_t* __restrict__ label) {
  extern __shared__ __attribute__((aligned(16))) float wsm[];  // [U][CMAX] then [CMAX] bias
  float* bsm = wsm + U * CMAX;
  for (int i = threadIdx.x; i < U * CMAX; i += 256) {
    const int u = i / CMAX, k = i - u * CMAX;
    wsm[i] = k < C ? ws_g[u * C + k] : 0.f;
  }
  if (threadIdx.x < CMAX) bsm[threadIdx.x] = threadIdx.x < C ? bs_g[threadIdx.x] : 0.f;
  __syncthreads();
  const int Ho = Hi * 8, Wo = Wi * 8;
  // block = 32 x 8 output pixels so a wave covers 2 rows x 32 columns (few distinct source pixels)
  const int tilesx = (Wo + 31) / 32;
  const int tx = blockIdx.x % tilesx;
  int r = blockIdx.x / tilesx;
  const int tilesy = (Ho + 7) / 8;
  const int ty = r % tilesy;
  const int n = r / tilesy;
  const int ox = tx * 32 + (threadIdx.x & 31), oy = ty * 8 + (threadIdx.x >> 5);
  if (ox >= Wo || oy >= Ho) return;
  int iy1, ix1;
  float wy1, wy0, wx1, wx0;
  bilinear_taps<8>(oy, iy1, wy1, wy0);
  bilinear_taps<8>(ox, ix1, wx1, wx0);
  const float w00 = wy0 * wx0, w01 = wy0 * wx1, w10 = wy1 * wx0, w11 = wy1 * wx1;
  const __bf16* p00 = f + (((int64_t)n * (Hi + 2) + iy1) * (Wi + 2) + ix1) * U;
  const int64_t rowp = (int64_t)(Wi + 2) * U;
  float sc[CMAX];
#pragma unroll
  for (int k = 0; k < CMAX; ++k) sc[k] = 0.f;
  for (int u0 = 0; u0 < U; u0 += 8) {
    const u32x4 a00 = *reinterpret_cast<const u32x4*>(p00 + u0), a01 = *reinterpret_cast<const u32x4*>(p00 + U + u0);
    const u32x4 a10 = *reinterpret_cast<const u32x4*>(p00 + rowp + u0),
                a11 = *reinterpret_cast<const u32x4*>(p00 + rowp + U + u0);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int sh = (i & 1) * 16, q = i >> 1;
      const float f00 = bf16_bits_to_f32((a00[q] >> sh) & 0xffffu), f01 = bf16_bits_to_f32((a01[q] >> sh) & 0xffffu);
      const float f10 = bf16_bits_to_f32((a10[q] >> sh) & 0xffffu), f11 = bf16_bits_to_f32((a11[q] >> sh) & 0xffffu);
      float up = f00 * w00 + f01 * w01 + f10 * w10 + f11 * w11;
      up = fmaxf(up, 0.f);
      const float* wrow = wsm + (u0 + i) * CMAX;
#pragma unroll
      for (int k4 = 0; k4 < CMAX; k4 += 4) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(wrow + k4);
        sc[k4] += up * wv.x;
        sc[k4 + 1] += up * wv.y;
        sc[k4 + 2] += up * wv.z;
        sc[k4 + 3] += up * wv.w;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < CMAX; ++k) sc[k] += bsm[k];
  const int64_t opix = ((int64_t)n * Ho + oy) * Wo + ox;
  if (score) {
#pragma unroll
    for (int k = 0; k < CMAX; ++k)
      if (k < C) score[opix * C + k] = sc[k];
  }
  if (prob || label) {
    float m = sc[0];
#pragma unroll
    for (int k = 1; k < CMAX; ++k)
      if (k < C) m = fmaxf(m, sc[k]);
    float e[CMAX];
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < CMAX; ++k) {
      e[k] = k < C ? expf(sc[k] - m) : 0.f;
      sum += e[k];
    }
    float best = -1.f;
    int bi = 0;
#pragma unroll
    for (int k = 0; k < CMAX; ++k) {
      const float p = e[k] / sum;
      if (k < C) {
        if (prob) prob[opix * C + k] = p;
        if (p > best) {
          best = p;
          bi = k;
        }
      }
    }
    if (label) label[opix] = bi;
  }
}

// ---- softmax + argmax on dense fp32 scores (basic_fusion_model.py:21-22) -------------------------
template <int CMAX>
__global__ __launch_bounds__(256) void softmax_argmax_kernel(const float* __restrict__ score, int64_t npix, int C,
                                                            float* __restrict__ prob, int64_t* __restrict__ label) {
  for (int64_t pix = (int64_t)blockIdx.x * 256 + threadIdx.x; pix < npix; pix += (int64_t)gridDim.x * 256) {
    float sc[CMAX];
#pragma unroll
    for (int k = 0; k < CMAX; ++k) sc[k] = k < C ? score[pix * C + k] : 0.f;
    float m = sc[0];
#pragma unroll
    for (int k = 1; k < CMAX; ++k)
      if (k < C) m = fmaxf(m, sc[k]);
    float e[CMAX];
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < CMAX; ++k) {
      e[k] = k < C ? expf(sc[k] - m) : 0.f;
      sum += e[k];
    }
    float best = -1.f;
    int bi = 0;
#pragma unroll
    for (int k = 0; k < CMAX; ++k) {
      const float p = e[k] / sum;
      if (k < C) {
        if (prob) prob[pix * C + k] = p;
        if (p > best) {
          best = p;
          bi = k;
        }
      }
    }
    if (label) label[pix] = bi;
  }
}

inline int grid_for(int64_t total, int per_block = 256, int cap = 8192) {
  int64_t g = (total + per_block - 1) / per_block;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

extern "C" int xv_conv2d_first_fwd(const float* x, int n, int h, int w, int cin, const float* w_hwio,
                                   const float* bias, const xv_act* y, int relu, void* stream) {
  XV_CHECK_ARG(x && w_hwio && bias && y && y->data);
  XV_CHECK_SHAPE(n > 0 && h > 0 && w > 0 && cin >= 1 && cin <= 4);
  XV_CHECK_SHAPE(y->n == n && y->h == h && y->w == w && y->c == 64);
  const int64_t npix = (int64_t)n * h * w;
  const unsigned grid = (unsigned)((npix + 255) / 256);
  hipStream_t s = (hipStream_t)stream;
  __bf16* yp = (__bf16*)y->data;
  switch (cin) {
    case 1: hipLaunchKernelGGL(conv_first_kernel<1>, dim3(grid), dim3(256), 0, s, x, w_hwio, bias, yp, n, h, w, relu); break;
    case 2: hipLaunchKernelGGL(conv_first_kernel<2>, dim3(grid), dim3(256), 0, s, x, w_hwio, bias, yp, n, h, w, relu); break;
    case 3: hipLaunchKernelGGL(conv_first_kernel<3>, dim3(grid), dim3(256), 0, s, x, w_hwio, bias, yp, n, h, w, relu); break;
    default: hipLaunchKernelGGL(conv_first_kernel<4>, dim3(grid), dim3(256), 0, s, x, w_hwio, bias, yp, n, h, w, relu); break;
  }
  return xv_launch_status();
}

extern "C" int xv_maxpool2x2_fwd(const xv_act* x, const xv_act* y, void* stream) {
  XV_CHECK_ARG(x && y && x->data && y->data);
  XV_CHECK_SHAPE(x->n > 0 && x->h > 0 && x->w > 0 && x->c > 0 && (x->c & 7) == 0 && (x->h & 1) == 0 && (x->w & 1) == 0);
  XV_CHECK_SHAPE(y->n == x->n && y->h == x->h / 2 && y->w == x->w / 2 && y->c == x->c);
  const int64_t total = (int64_t)y->n * y->h * y->w * (y->c >> 3);
  hipLaunchKernelGGL(maxpool_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const __bf16*)x->data,
                     (__bf16*)y->data, y->n, y->h, y->w, y->c);
  return xv_launch_status();
}

extern "C" int xv_upsample2x_relu_add(const xv_act* x, const xv_act* residual, const xv_act* y, void* stream) {
  XV_CHECK_ARG(x && y && x->data && y->data);
  XV_CHECK_SHAPE(x->n > 0 && x->h > 0 && x->w > 0 && x->c > 0 && (x->c & 7) == 0);
  XV_CHECK_SHAPE(y->n == x->n && y->h == 2 * x->h && y->w == 2 * x->w && y->c == x->c);
  const __bf16* res = nullptr;
  if (residual && residual->data) {
    XV_CHECK_SHAPE(residual->n == y->n && residual->h == y->h && residual->w == y->w && residual->c == y->c);
    res = (const __bf16*)residual->data;
  }
  const int64_t total = (int64_t)y->n * y->h * y->w * (y->c >> 3);
  hipLaunchKernelGGL(upsample2x_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                     (const __bf16*)x->data, res, (__bf16*)y->data, x->n, x->h, x->w, x->c);
  return xv_launch_status();
}

extern "C" int xv_decoder_head_fwd(const xv_act* fused, const float* w_score, const float* b_score, int num_classes,
                                   float* score, float* prob, int64_t* label, void* stream) {
  XV_CHECK_ARG(fused && fused->data && w_score && b_score);
  XV_CHECK_ARG(score || prob || label);
  XV_CHECK_SHAPE(fused->n > 0 && fused->h > 0 && fused->w > 0);
  XV_CHECK_SHAPE(fused->c > 0 && (fused->c & 7) == 0 && fused->c <= 128 && num_classes >= 1 && num_classes <= 32);
  const int Ho = fused->h * 8, Wo = fused->w * 8;
  const int64_t nblk = (int64_t)((Wo + 31) / 32) * ((Ho + 7) / 8) * fused->n;
  XV_CHECK_SHAPE(nblk <= 0x7fffffff);
  hipStream_t s = (hipStream_t)stream;
  const __bf16* f = (const __bf16*)fused->data;
  if (num_classes <= 16) {
    const size_t lds = (size_t)(fused->c * 16 + 16) * 4;
    hipLaunchKernelGGL(decoder_head_kernel<16>, dim3((unsigned)nblk), dim3(256), lds, s, f, w_score, b_score, fused->n,
                       fused->h, fused->w, fused->c, num_classes, score, prob, label);
  } else {
    const size_t lds = (size_t)(fused->c * 32 + 32) * 4;
    hipLaunchKernelGGL(decoder_head_kernel<32>, dim3((unsigned)nblk), dim3(256), lds, s, f, w_score, b_score, fused->n,
                       fused->h, fused->w, fused->c, num_classes, score, prob, label);
  }
  return xv_launch_status();
}

extern "C" int xv_softmax_argmax(const float* score, int64_t npix, int num_classes, float* prob, int64_t* label,
                                 void* stream) {
  XV_CHECK_ARG(score && (prob || label));
  XV_CHECK_SHAPE(npix > 0 && num_classes >= 1 && num_classes <= 32);
  hipStream_t s = (hipStream_t)stream;
  if (num_classes <= 16)
    hipLaunchKernelGGL(softmax_argmax_kernel<16>, dim3(grid_for(npix)), dim3(256), 0, s, score, npix, num_classes, prob,
                       label);
  else
    hipLaunchKernelGGL(softmax_argmax_kernel<32>, dim3(grid_for(npix)), dim3(256), 0, s, score, npix, num_classes, prob,
                       label);
  return xv_launch_status();
}
