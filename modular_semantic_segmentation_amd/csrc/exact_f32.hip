// The FCN trunk in plain fp32 ("exact" mode, conv_dtype='fp32'): the reference graph's own arithmetic type
// (tf.layers.conv2d / max_pooling2d / conv2d_transpose on float32, simple_fcn.py:39-87, custom_layers.py:71-139) on dense
// unpadded NHWC float32 maps, with fp32 FMAs and no bf16 storage anywhere.  It exists for the parity contract, not for
// speed (about 1/100 of the MFMA path): on the same weights its label maps must equal the fp32 oracle's, which turns
// "the bf16 path agrees with the oracle on clear margins" into evidence that the pixels the bf16 path loses are lost to
// bf16 storage and not to a kernel (tests/test_exact_f32_gpu.py).
//
//   xv_conv2d_f32        3x3 'same' / 1x1, bias, optional relu, optional fused 2x2 max-pool output
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
__global__ __launch_bounds__(256) void upsample2x_f32_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                            float* __restrict__ y, int N, int Hi, int Wi, int C) {
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
    y[i] = fmaxf(a, 0.f) + (res != nullptr ? res[i] : 0.f);
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

extern "C" int xv_conv2d_f32(const float* x, int n, int h, int w, int cin, const float* w_hwio, const float* bias, int k,
                             int cout, int relu, float* y, void* stream) {
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
  hipLaunchKernelGGL(upsample2x_f32_kernel, dim3(f32_grid(total)), dim3(256), 0, (hipStream_t)stream, x, residual, y, n, h, w, c);
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
