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
#include "xv_common.h"

namespace {

struct ConvArgs {
  const __bf16* x;
  const __bf16* wpk;
  const float* bias;
  __bf16* y;       // may be null
  __bf16* pooled;  // may be null
  int N, H, W, Cin, Cout;
  int tiles_x, tiles_y, n_ct;
  int relu;
};

template <int MT, int WR, int WC, int NW, int KS>
struct ConvCfg {
  static constexpr int NT = 64 * WR * WC * NW;
  static constexpr int TH = MT * WR, TW = 16 * WC;
  static constexpr int HALO = (KS == 3) ? 1 : 0;
  static constexpr int HH = TH + 2 * HALO, HW = TW + 2 * HALO;
  static constexpr int NPIX = HH * HW;
  static constexpr int BN = 64 * NW;
  static constexpr int A_BYTES = ((NPIX * 128 + 255) / 256) * 256;
  static constexpr int B_BYTES = BN * 128;
  static constexpr int NTAPS = KS * KS;
  static constexpr int LDS_BYTES = A_BYTES + 2 * B_BYTES;
  static constexpr int A_ITERS = (NPIX * 8 + NT - 1) / NT;
  static constexpr int B_ITERS = B_BYTES / 16 / NT;
  static_assert(B_BYTES % (16 * NT) == 0, "weight tile must split evenly over the threads");
};

// OCC = waves per SIMD the register allocation is bounded for (2 -> <= 256 VGPRs, two 4-wave
// workgroups per CU; 1 -> up to 512).  The next-chunk patch prefetch (PFA) keeps A_ITERS*4 extra
// registers live under the last tap, which only fits with MT = 4 or OCC = 1.
template <int MT, int WR, int WC, int NW, int KS, int OCC>
__global__ __launch_bounds__(64 * WR * WC * NW, OCC * 4 / (WR * WC * NW) > 0 ? OCC * 4 / (WR * WC * NW) : 1) void conv_mfma_kernel(ConvArgs a) {
  using C = ConvCfg<MT, WR, WC, NW, KS>;
  constexpr bool PFA = (MT == 4) || (OCC == 1);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* As = smem;
  char* Bs = smem + C::A_BYTES;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave % NW;
  const int wm = wave / NW;
  const int wr = wm / WC;
  const int wc = wm % WC;

  // XCD-aware block remap (bijective): consecutive logical ids share an XCD (and its L2), and
  // consecutive logical ids are the cout tiles of ONE pixel patch, then the neighbouring patch.
  const int nblk = gridDim.x;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, q = nblk >> 3, r = nblk & 7;
  const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  const int ct = lid % a.n_ct;
  int t = lid / a.n_ct;
  const int tx = t % a.tiles_x;
  t /= a.tiles_x;
  const int ty = t % a.tiles_y;
  const int n = t / a.tiles_y;

  const int y0 = ty * C::TH, x0 = tx * C::TW;  // logical coords of the patch origin
  const int co0 = ct * C::BN;
  const int H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout;
  const int Wp = W + 2;
  const __bf16* ximg = a.x + (int64_t)n * (H + 2) * Wp * Cin;
  const int nchunks = Cin >> 6;

  f32x4 acc[MT][4];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int l15 = lane & 15, lg = lane >> 4;
  const int pbase = (wr * MT) * C::HW + wc * 16 + l15;  // halo-patch pixel of (row 0 of this wave, tap (0,0))
  const int wrow = wn * 64 + l15;                       // weight-tile row of n-tile 0
  const int wswz = (l15 >> 1) & 7;                      // swizzle term of every weight row this lane reads

  // staging helpers: global -> registers (issued early) and registers -> LDS (after the barrier).
  // Patch coordinates are clamped to the zero border of the padded buffer, so every load is
  // in-bounds and unconditional; outputs fed by clamped pixels are never stored.
  auto a_load = [&](int chunk, auto& v) {
#pragma unroll
    for (int it = 0; it < C::A_ITERS; ++it) {
      int idx = tid + it * C::NT;
      idx = idx < C::NPIX * 8 ? idx : C::NPIX * 8 - 1;
      const int p = idx >> 3, s = idx & 7;
      const int hy = p / C::HW, hx = p - hy * C::HW;
      int yy = y0 + hy + (1 - C::HALO), xx = x0 + hx + (1 - C::HALO);  // padded coords
      yy = yy < H + 1 ? yy : H + 1;
      xx = xx < W + 1 ? xx : W + 1;
      v[it] = *reinterpret_cast<const u32x4*>(ximg + ((int64_t)yy * Wp + xx) * Cin + chunk * 64 + s * 8);
    }
  };
  auto a_store = [&](const auto& v) {
#pragma unroll
    for (int it = 0; it < C::A_ITERS; ++it) {
      const int idx = tid + it * C::NT;
      const int p = idx >> 3, s = idx & 7;
      if (idx < C::NPIX * 8) *reinterpret_cast<u32x4*>(As + p * 128 + (xv_swz(p, s) << 4)) = v[it];
    }
  };
  auto b_load = [&](int tap, int chunk, u32x4(&v)[C::B_ITERS]) {
    const char* src = reinterpret_cast<const char*>(a.wpk) + (((int64_t)(tap * nchunks + chunk) * Cout + co0) << 7);
#pragma unroll
    for (int it = 0; it < C::B_ITERS; ++it) v[it] = *reinterpret_cast<const u32x4*>(src + ((tid + it * C::NT) << 4));
  };
  auto b_store = [&](int buf, const u32x4(&v)[C::B_ITERS]) {
    char* dst = Bs + buf * C::B_BYTES;
#pragma unroll
    for (int it = 0; it < C::B_ITERS; ++it) *reinterpret_cast<u32x4*>(dst + ((tid + it * C::NT) << 4)) = v[it];
  };

  u32x4 areg[PFA ? C::A_ITERS : 1];
  u32x4 breg[C::B_ITERS];
  if constexpr (PFA) a_load(0, areg);
  b_load(0, 0, breg);
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    if (chunk > 0) __syncthreads();  // all waves are done reading As / Bs of the previous chunk
    if constexpr (PFA) {
      a_store(areg);
    } else {
      u32x4 atmp[C::A_ITERS];
      a_load(chunk, atmp);
      a_store(atmp);
    }
    b_store(0, breg);
    __syncthreads();

    for (int tap = 0; tap < C::NTAPS; ++tap) {
      const int cur = tap & 1;
      const char* Bcur = Bs + cur * C::B_BYTES;
      const bool more = tap + 1 < C::NTAPS;
      if (more) {
        b_load(tap + 1, chunk, breg);
      } else if (chunk + 1 < nchunks) {
        if constexpr (PFA) a_load(chunk + 1, areg);  // next chunk's patch + first weight tile travel under the last tap's MFMAs
        b_load(0, chunk + 1, breg);
      }
      const int dy = (KS == 3) ? tap / 3 : 0;
      const int dx = (KS == 3) ? tap - dy * 3 : 0;
      const int ptap = pbase + dy * C::HW + dx;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int slot = kk * 4 + lg;
        bf16x8 wf[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
          wf[j] = *reinterpret_cast<const bf16x8*>(Bcur + (wrow + j * 16) * 128 + ((slot ^ wswz) << 4));
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const int p = ptap + i * C::HW;
          const bf16x8 xf = *reinterpret_cast<const bf16x8*>(As + p * 128 + (xv_swz(p, slot) << 4));
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf, acc[i][j], 0, 0, 0);
        }
      }
      if (more) {
        b_store(cur ^ 1, breg);
        __syncthreads();
      }
    }
  }

  // ---- epilogue: bias + relu, bf16, 8-byte NHWC stores (4 consecutive channels per lane) --------
  const int px = x0 + wc * 16 + l15;
  const int cbase = co0 + wn * 64 + lg * 4;
  f32x4 bj[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) bj[j] = *reinterpret_cast<const f32x4*>(a.bias + cbase + j * 16);
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
    __bf16* yimg = a.y + (int64_t)n * (H + 2) * Wp * Cout;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int py = y0 + wr * MT + i;
      if (py < H && px < W) {
        __bf16* dst = yimg + ((int64_t)(py + 1) * Wp + (px + 1)) * Cout + cbase;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x4 v = acc[i][j];
          *reinterpret_cast<u32x2*>(dst + j * 16) = u32x2{pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w)};
        }
      }
    }
  }
  if (a.pooled != nullptr) {
    // fused max_pooling2d(2,2): rows (i, i+1) live in this lane, columns (px, px^1) in lanes l, l^1
    const int Hq = H >> 1, Wq = W >> 1;
    __bf16* qimg = a.pooled + (int64_t)n * (Hq + 2) * (Wq + 2) * Cout;
#pragma unroll
    for (int i = 0; i < MT; i += 2) {
      const int py = y0 + wr * MT + i;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x4 m;
        m.x = fmaxf(acc[i][j].x, acc[i + 1][j].x);
        m.y = fmaxf(acc[i][j].y, acc[i + 1][j].y);
        m.z = fmaxf(acc[i][j].z, acc[i + 1][j].z);
        m.w = fmaxf(acc[i][j].w, acc[i + 1][j].w);
        m.x = fmaxf(m.x, __shfl_xor(m.x, 1));
        m.y = fmaxf(m.y, __shfl_xor(m.y, 1));
        m.z = fmaxf(m.z, __shfl_xor(m.z, 1));
        m.w = fmaxf(m.w, __shfl_xor(m.w, 1));
        if ((lane & 1) == 0 && py < H && px < W) {
          __bf16* dst = qimg + ((int64_t)((py >> 1) + 1) * (Wq + 2) + ((px >> 1) + 1)) * Cout + cbase + j * 16;
          *reinterpret_cast<u32x2*>(dst) = u32x2{pack_bf16x2(m.x, m.y), pack_bf16x2(m.z, m.w)};
        }
      }
    }
  }
}

template <int MT, int WR, int WC, int NW, int KS, int OCC>
int launch_conv(const ConvArgs& a0, hipStream_t stream) {
  using C = ConvCfg<MT, WR, WC, NW, KS>;
  ConvArgs a = a0;
  a.tiles_x = (a.W + C::TW - 1) / C::TW;
  a.tiles_y = (a.H + C::TH - 1) / C::TH;
  a.n_ct = a.Cout / C::BN;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_kernel<MT, WR, WC, NW, KS, OCC>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const int64_t nblk = (int64_t)a.tiles_x * a.tiles_y * a.N * a.n_ct;
  if (nblk <= 0 || nblk > 0x7fffffff) return XV_ESHAPE;
  if (a.pooled && (C::TH & 1)) return XV_ESHAPE;
  hipLaunchKernelGGL((conv_mfma_kernel<MT, WR, WC, NW, KS, OCC>), dim3((unsigned)nblk), dim3(C::NT), C::LDS_BYTES, stream, a);
  return xv_launch_status();
}

// ---- weight packing: fp32 HWIO -> bf16 [tap][cin/64][cout][64], 16-byte slots swizzled -----------
__global__ void pack_weights_kernel(const float* __restrict__ w, __bf16* __restrict__ out, int taps, int cin,
                                    int cout) {
  const int64_t total = (int64_t)taps * cin * cout;
  const int nchunks = cin >> 6;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    // destination-linear index: [tap][chunk][co][phys_slot][e]
    const int e = (int)(idx & 7);
    const int ps = (int)((idx >> 3) & 7);
    int64_t rest = idx >> 6;
    const int co = (int)(rest % cout);
    rest /= cout;
    const int chunk = (int)(rest % nchunks);
    const int tap = (int)(rest / nchunks);
    const int s = xv_swz(co, ps);  // involution: logical slot stored at this physical slot
    const int ci = chunk * 64 + s * 8 + e;
    out[idx] = (__bf16)w[((int64_t)tap * cin + ci) * cout + co];
  }
}

// ---- tile configurations ---------------------------------------------------------------------
// id: pixel patch x output channels, waves, LDS, workgroups per CU (register bound)
//   0: 16x16 x 128, 4 waves, 74 KB, 2/CU          1:  8x16 x 128, 4 waves, 55 KB, 2/CU
//   2:  8x32 x 128, 4 waves, 75 KB, 2/CU          3: 16x32 x 128, 8 waves, 110 KB, 1/CU
//   4: 16x16 x  64, 4 waves, 57 KB, 2/CU          5: 16x32 x  64, 4 waves, 94 KB, 1/CU
//   6:  8x32 x  64, 4 waves, 59 KB, 2/CU          7:  8x16 x 256, 4 waves, 87 KB, 1/CU
//   8: as 0 but 1/CU with the patch prefetch      9: as 2 but 1/CU with the patch prefetch
constexpr int XV_NUM_CONV_CFG = 10;
struct Geo {
  int th, tw, bn, per_cu;
};
const Geo kGeo[XV_NUM_CONV_CFG] = {{16, 16, 128, 2}, {8, 16, 128, 2}, {8, 32, 128, 2}, {16, 32, 128, 1},
                                   {16, 16, 64, 2},  {16, 32, 64, 1}, {8, 32, 64, 2},  {8, 16, 256, 1},
                                   {16, 16, 128, 1}, {8, 32, 128, 1}};

template <int KS>
int launch_cfg(int cfg, const ConvArgs& a, hipStream_t s) {
  if (cfg < 0 || cfg >= XV_NUM_CONV_CFG) return XV_EINVAL;
  if (a.Cout % kGeo[cfg].bn) return XV_ESHAPE;
  switch (cfg) {
    case 0: return launch_conv<8, 2, 1, 2, KS, 2>(a, s);
    case 1: return launch_conv<4, 2, 1, 2, KS, 2>(a, s);
    case 2: return launch_conv<8, 1, 2, 2, KS, 2>(a, s);
    case 3: return launch_conv<8, 2, 2, 2, KS, 2>(a, s);
    case 4: return launch_conv<4, 4, 1, 1, KS, 2>(a, s);
    case 5: return launch_conv<8, 2, 2, 1, KS, 1>(a, s);
    case 6: return launch_conv<4, 2, 2, 1, KS, 2>(a, s);
    case 7: return launch_conv<8, 1, 1, 4, KS, 1>(a, s);
    case 8: return launch_conv<8, 2, 1, 2, KS, 1>(a, s);
    default: return launch_conv<8, 1, 2, 2, KS, 1>(a, s);
  }
}

// Default choice: fewest "rounds" of workgroups over the 256 CUs, ties to the larger tile.
int pick_cfg(const ConvArgs& a) {
  int best = -1;
  double best_cost = 0;
  for (int c = 0; c < 8; ++c) {
    const Geo& g = kGeo[c];
    if (a.Cout % g.bn) continue;
    if (a.pooled && (g.th & 1)) continue;
    const double blocks = (double)((a.H + g.th - 1) / g.th) * ((a.W + g.tw - 1) / g.tw) * a.N * (a.Cout / g.bn);
    const double slots = 256.0 * g.per_cu;
    const double rounds = __builtin_ceil(blocks / slots);
    // time ~ rounds * work per workgroup * (co-resident workgroups share the CU's MFMA pipes)
    double cost = rounds * (double)g.th * g.tw * g.bn * g.per_cu;
    // halo + weight re-fetch overhead favours the larger tiles at equal rounds
    cost *= 1.0 + 0.08 * (256.0 * 128.0) / ((double)g.th * g.tw * g.bn);
    if (best < 0 || cost < best_cost) {
      best = c;
      best_cost = cost;
    }
  }
  return best;
}

int conv_fwd_impl(const xv_act* x, const void* w_packed, const float* bias, const xv_act* y, const xv_act* pooled,
                  int k, int relu, int cfg, void* stream) {
  XV_CHECK_ARG(x && x->data && w_packed && bias && y);
  XV_CHECK_ARG(y->data || (pooled && pooled->data));
  XV_CHECK_SHAPE(k == 1 || k == 3);
  XV_CHECK_SHAPE(x->n > 0 && x->h > 0 && x->w > 0 && x->c > 0 && (x->c & 63) == 0);
  XV_CHECK_SHAPE(y->n == x->n && y->h == x->h && y->w == x->w && y->c > 0 && (y->c & 63) == 0);
  XV_CHECK_ARG((((uintptr_t)x->data | (uintptr_t)w_packed | (uintptr_t)bias | (uintptr_t)y->data) & 15) == 0);
  ConvArgs a{};
  a.x = (const __bf16*)x->data;
  a.wpk = (const __bf16*)w_packed;
  a.bias = bias;
  a.y = (__bf16*)y->data;
  a.pooled = nullptr;
  a.N = x->n;
  a.H = x->h;
  a.W = x->w;
  a.Cin = x->c;
  a.Cout = y->c;
  a.relu = relu;
  if (pooled && pooled->data) {
    XV_CHECK_SHAPE(k == 3 && (x->h & 1) == 0 && (x->w & 1) == 0);
    XV_CHECK_SHAPE(pooled->n == x->n && pooled->h == x->h / 2 && pooled->w == x->w / 2 && pooled->c == y->c);
    XV_CHECK_ARG(((uintptr_t)pooled->data & 15) == 0);
    a.pooled = (__bf16*)pooled->data;
  }
  if (cfg < 0) cfg = pick_cfg(a);
  if (cfg < 0) return XV_ESHAPE;
  hipStream_t s = (hipStream_t)stream;
  return k == 3 ? launch_cfg<3>(cfg, a, s) : launch_cfg<1>(cfg, a, s);
}

}  // namespace

extern "C" size_t xv_packed_weight_bytes(int k, int cin, int cout) {
  if ((k != 1 && k != 3) || cin <= 0 || cout <= 0 || (cin & 63) || (cout & 63)) return 0;
  return (size_t)k * k * cin * cout * 2;
}

extern "C" int xv_pack_conv_weights(const float* w_hwio, void* packed, int k, int cin, int cout, void* stream) {
  XV_CHECK_ARG(w_hwio && packed);
  XV_CHECK_SHAPE((k == 1 || k == 3) && cin > 0 && cout > 0 && (cin & 63) == 0 && (cout & 63) == 0);
  const int64_t total = (int64_t)k * k * cin * cout;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_hwio, (__bf16*)packed,
                     k * k, cin, cout);
  return xv_launch_status();
}

extern "C" int xv_conv2d_fwd(const xv_act* x, const void* w_packed, const float* bias, const xv_act* y,
                             const xv_act* pooled, int k, int relu, void* stream) {
  return conv_fwd_impl(x, w_packed, bias, y, pooled, k, relu, -1, stream);
}

extern "C" int xv_conv2d_fwd_cfg(const xv_act* x, const void* w_packed, const float* bias, const xv_act* y,
                                 const xv_act* pooled, int k, int relu, int cfg, void* stream) {
  return conv_fwd_impl(x, w_packed, bias, y, pooled, k, relu, cfg, stream);
}

extern "C" int xv_conv2d_num_cfgs(void) { return XV_NUM_CONV_CFG; }
