// Per-pixel probabilistic fusion, sufficient statistics and confusion matrix for gfx950.
// All HBM-bound: one thread per pixel, class vectors in registers, tables in LDS.
#include <stdint.h>
#include <stdlib.h>

#include "xv_common.h"

// Floating-point contraction by the SOURCE only (a * b + c inside one expression), never across statements: the fused
// two-expert head and the unfused path (decoder head -> probability maps -> fusion kernel) must produce the same bits, and
// under the default -ffp-contract=fast the optimizer fuses a product into a later sum wherever the two happen to meet --
// round 4: specialising the fused head on the class count removed a select between `p = e * rsum` and `sum += p`, the
// compiler made it an fma there and not in the kernel that reads p back from memory, and one pixel in a million flipped.
#pragma clang fp contract(on)

namespace {

constexpr int MAXE = 4;

struct LabelPtrs {
  const int64_t* p[MAXE];
};
struct ProbPtrs {
  const float* p[MAXE];
};

// A pixel's C probabilities (contiguous floats) into registers: 16-byte loads where the class count is a multiple of four
// (every row is then 16-byte aligned; the base pointers are checked by the host) -- three requests per pixel at C = 12
// instead of twelve, each of which touched the same dozen cache lines of the wave's 3 KB span.  Slots past C read as 0.
template <int CMAX>
__device__ static __forceinline__ void load_row(const float* __restrict__ p, int C, bool vec, float (&x)[CMAX]) {
  if (vec) {
#pragma unroll
    for (int q = 0; q < CMAX / 4; ++q) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (4 * q < C) v = *reinterpret_cast<const f32x4*>(p + 4 * q);
      x[4 * q] = v.x, x[4 * q + 1] = v.y, x[4 * q + 2] = v.z, x[4 * q + 3] = v.w;
    }
  } else {
#pragma unroll
    for (int k = 0; k < CMAX; ++k) x[k] = k < C ? p[k] : 0.f;
  }
}

inline int grid_for(int64_t total, int per_block = 256, int cap = 8192) {
  int64_t g = (total + per_block - 1) / per_block;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

// bayes_mix.py:12-58 + :161.  score[c] = (ll_0[c] + ll_1[c] + ...) + logprior[c]; argmax, first index on ties.
template <int CMAX>
__global__ __launch_bounds__(256) void bayes_fuse_kernel(LabelPtrs labels, int E, const float* __restrict__ loglik,
                                                        const float* __restrict__ logprior, int C, int64_t npix,
                                                        int64_t* __restrict__ fused, float* __restrict__ score_out) {
  extern __shared__ __attribute__((aligned(16))) float tab[];  // [E][C][CMAX] then [CMAX]
  float* lp = tab + E * C * CMAX;
  for (int i = threadIdx.x; i < E * C * CMAX; i += 256) {
    const int k = i % CMAX, row = i / CMAX;
    tab[i] = k < C ? loglik[row * C + k] : 0.f;
  }
  if (threadIdx.x < CMAX) lp[threadIdx.x] = threadIdx.x < C ? logprior[threadIdx.x] : 0.f;
  __syncthreads();
  for (int64_t pix = (int64_t)blockIdx.x * 256 + threadIdx.x; pix < npix; pix += (int64_t)gridDim.x * 256) {
    float sc[CMAX];
#pragma unroll
    for (int k = 0; k < CMAX; ++k) sc[k] = 0.f;
    for (int e = 0; e < E; ++e) {
      int64_t l = labels.p[e][pix];
      l = l < 0 ? 0 : (l >= C ? C - 1 : l);  // tf.gather would fault; clamp keeps the kernel memory-safe
      const float* row = tab + ((int64_t)e * C + l) * CMAX;
#pragma unroll
      for (int k = 0; k < CMAX; ++k) sc[k] = e == 0 ? row[k] : sc[k] + row[k];
    }
    float best = 0.f;
    int bi = 0;
#pragma unroll
    for (int k = 0; k < CMAX; ++k) {
      const float v = sc[k] + lp[k];
      if (k < C) {
        if (score_out) score_out[pix * C + k] = v;
        if (k == 0 || v > best) {
          best = v;
          bi = k;
        }
      }
    }
    fused[pix] = bi;
  }
}

// The same decision for TWO experts and no score output through a [C][C] table built by the workgroup itself: entry (a, b)
// is the argmax the per-pixel kernel above computes for the label pair (a, b) -- the same sums in the same order, so the same
// label bit for bit -- and a pixel is then two label loads, one LDS lookup and a store (the per-pixel form read 2 x 64 bytes
// of table rows per pixel: 0.56 of HBM against the lookup kernel's 0.75).
template <int CMAX>
__global__ __launch_bounds__(256) void bayes_fuse2_kernel(const int64_t* __restrict__ la, const int64_t* __restrict__ lb,
                                                         const float* __restrict__ loglik, const float* __restrict__ logprior,
                                                         int C, int64_t npix, int64_t* __restrict__ fused) {
  extern __shared__ int dec[];           // [C][C] decisions, then the two [C][C] tables and the prior as floats
  float* tab = reinterpret_cast<float*>(dec + C * C);
  for (int i = threadIdx.x; i < 2 * C * C + C; i += 256) tab[i] = i < 2 * C * C ? loglik[i] : logprior[i - 2 * C * C];
  __syncthreads();
  for (int i = threadIdx.x; i < C * C; i += 256) {
    const float* ra = tab + (i / C) * C;
    const float* rb = tab + (C + i % C) * C;
    const float* lp = tab + 2 * C * C;
    float best = 0.f;
    int bi = 0;
    for (int k = 0; k < C; ++k) {
      float sc = ra[k];
      sc = sc + rb[k];
      const float v = sc + lp[k];
      if (k == 0 || v > best) {
        best = v;
        bi = k;
      }
    }
    dec[i] = bi;
  }
  __syncthreads();
  typedef long long i64x2 __attribute__((ext_vector_type(2)));
  auto clampc = [&](long long l) { return (int)(l < 0 ? 0 : (l >= C ? C - 1 : l)); };
  const int64_t pairs = npix >> 1;
  for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < pairs; q += (int64_t)gridDim.x * 256) {
    const i64x2 a = *reinterpret_cast<const i64x2*>(la + 2 * q), b = *reinterpret_cast<const i64x2*>(lb + 2 * q);
    const i64x2 f = {dec[clampc(a.x) * C + clampc(b.x)], dec[clampc(a.y) * C + clampc(b.y)]};
    *reinterpret_cast<i64x2*>(fused + 2 * q) = f;
  }
  if ((npix & 1) && blockIdx.x == 0 && threadIdx.x == 0)
    fused[npix - 1] = dec[clampc(la[npix - 1]) * C + clampc(lb[npix - 1])];
}

// bayes_mix.py:61-112 / experiments/timing.py:87-115: fused = lut[a][b]
__global__ __launch_bounds__(256) void bayes_lut_kernel(const int64_t* __restrict__ a, const int64_t* __restrict__ b,
                                                       const int64_t* __restrict__ lut, int C, int64_t npix,
                                                       int64_t* __restrict__ fused) {
  extern __shared__ int lut_s[];
  for (int i = threadIdx.x; i < C * C; i += 256) lut_s[i] = (int)lut[i];
  __syncthreads();
  for (int64_t pix = (int64_t)blockIdx.x * 256 + threadIdx.x; pix < npix; pix += (int64_t)gridDim.x * 256) {
    int64_t la = a[pix], lb = b[pix];
    la = la < 0 ? 0 : (la >= C ? C - 1 : la);
    lb = lb < 0 ? 0 : (lb >= C ? C - 1 : lb);
    fused[pix] = lut_s[la * C + lb];
  }
}

// dirichlet_mix.py:14-36,96-136.
// EXACT: the class count IS the template parameter (12, the reference's): the class loop unrolls and the per-class sums sit
// in statically indexed registers (with a run-time count every class costs CMAX selects: 192 of the ~800 instructions per
// pixel); the dot products run over exactly C terms -- the same fmaf chain as the padded form, whose extra terms add 0 * 0.
template <int CMAX, bool EXACT = false>
__global__ __launch_bounds__(256) void dirichlet_fuse_kernel(ProbPtrs probs, int E, const float* __restrict__ am1,
                                                            const float* __restrict__ lognorm,
                                                            const float* __restrict__ logprior, int C_, int64_t npix,
                                                            int64_t* __restrict__ fused, float* __restrict__ score_out,
                                                            int vec) {
  const int C = EXACT ? CMAX : C_;
  extern __shared__ __attribute__((aligned(16))) float tab[];  // am1 [E][C][CMAX], lognorm [E][CMAX], logprior [CMAX]
  float* ln = tab + E * C * CMAX;
  float* lp = ln + E * CMAX;
  for (int i = threadIdx.x; i < E * C * CMAX; i += 256) {
    const int k = i % CMAX, row = i / CMAX;
    tab[i] = k < C ? am1[row * C + k] : 0.f;
  }
  for (int i = threadIdx.x; i < E * CMAX; i += 256) {
    const int k = i % CMAX, e = i / CMAX;
    ln[i] = k < C ? lognorm[e * C + k] : 0.f;
  }
  if (threadIdx.x < CMAX) lp[threadIdx.x] = threadIdx.x < C ? logprior[threadIdx.x] : 0.f;
  __syncthreads();
  for (int64_t pix = (int64_t)blockIdx.x * 256 + threadIdx.x; pix < npix; pix += (int64_t)gridDim.x * 256) {
    float total[CMAX];
#pragma unroll
    for (int c = 0; c < CMAX; ++c) total[c] = 0.f;
    // two experts (the usual case): both rows are requested before the first one's arithmetic starts
    float pre[2][CMAX];
    if (E == 2) {
      load_row<CMAX>(probs.p[0] + pix * C, C, vec, pre[0]);
      load_row<CMAX>(probs.p[1] + pix * C, C, vec, pre[1]);
    }
    for (int e = 0; e < E; ++e) {
      float lx[CMAX];
      if (E == 2) {
#pragma unroll
        for (int k = 0; k < CMAX; ++k) lx[k] = e == 0 ? pre[0][k] : pre[1][k];
      } else {
        load_row<CMAX>(probs.p[e] + pix * C, C, vec, lx);
      }
      float sum = 0.f;
#pragma unroll
      for (int k = 0; k < CMAX; ++k) sum += lx[k];
      const float rs = xv_fast_rcp(sum);  // (the same helper forms as the fused head: xv_common.h)
#pragma unroll
      for (int k = 0; k < CMAX; ++k) lx[k] = k < C ? xv_fast_log(1e-20f + lx[k] * rs) : 0.f;  // renormalise, then log(1e-20 + p)
      if (EXACT) {
#pragma unroll
        for (int c = 0; c < CMAX; ++c) {
          const float* row = tab + (e * CMAX + c) * CMAX;
          float dot = 0.f;
#pragma unroll
          for (int k = 0; k < CMAX; ++k) dot = fmaf(row[k], lx[k], dot);
          const float L = dot - ln[e * CMAX + c];
          total[c] = e == 0 ? L : total[c] + L;
        }
      } else {
        for (int c = 0; c < C; ++c) {
          const float* row = tab + ((int64_t)e * C + c) * CMAX;
          // explicit fmaf chain: the fused two-expert head (pointwise.hip fused_head_kernel) repeats this arithmetic and
          // must produce the same bits, so nothing is left to the compiler's contraction choices
          float dot = 0.f;
#pragma unroll
          for (int k = 0; k < CMAX; ++k) dot = fmaf(row[k], lx[k], dot);
          const float L = dot - ln[e * CMAX + c];
          // static register indexing: select instead of total[c] with runtime c
#pragma unroll
          for (int cc = 0; cc < CMAX; ++cc)
            if (cc == c) total[cc] = e == 0 ? L : total[cc] + L;
        }
      }
    }
    float best = 0.f;
    int bi = 0;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
      const float v = total[c] + lp[c];
      if (c < C) {
        if (score_out) score_out[pix * C + c] = v;
        if (c == 0 || v > best) {
          best = v;
          bi = c;
        }
      }
    }
    fused[pix] = bi;
  }
}

// dirichlet_fuse_kernel<CM, true> for two experts on PACKED fp32: the table in LDS transposed ([e][k][c] = alpha_e[c][k] - 1),
// so that the CM dot products of a pixel advance together, two classes per v_pk_fma_f32, each still the fmaf chain over k
// of the scalar form; the renormalisation fma and the ln 2 product two classes per instruction as well -- the same IEEE
// operations on the same operands, the sums in the same order: the same labels and scores bit for bit.  One pixel per
// thread and NO grid-stride loop: around a loop the compiler keeps the whole loop-invariant table in registers (256 VGPRs +
// 116 AGPRs, one wave per SIMD: 161 us).  16 images of 768x384 (tools/dirichlet_head_ab.py): scalar form 108-117 us (0.52-0.57
// of 8 TB/s, bound by its 288 scalar FMAs and their index arithmetic), this one 87 us (0.70; 5.65 TB/s of the 6.29 TB/s a
// float4 copy reaches); two / four pixels per thread sharing the table rows read from LDS: 89 / 92 us -- the rows are not
// what binds this kernel once the FMAs are packed, unlike the fused head (pointwise.hip).
template <int CM>
__global__ __launch_bounds__(256) void dirichlet_fuse_pk_kernel(const float* __restrict__ pa, const float* __restrict__ pb,
                                                               const float* __restrict__ am1, const float* __restrict__ lognorm,
                                                               const float* __restrict__ logprior, int64_t npix,
                                                               int64_t* __restrict__ fused, float* __restrict__ score_out) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  constexpr int H2 = CM / 2;
  extern __shared__ __attribute__((aligned(16))) float tab[];
  float* ln = tab + 2 * CM * CM;
  float* lp = ln + 2 * CM;
  for (int i = threadIdx.x; i < 2 * CM * CM; i += 256) {
    const int c = i % CM, k = (i / CM) % CM, e = i / (CM * CM);
    tab[i] = am1[(e * CM + c) * CM + k];
  }
  for (int i = threadIdx.x; i < 2 * CM; i += 256) ln[i] = lognorm[i];
  if (threadIdx.x < CM) lp[threadIdx.x] = logprior[threadIdx.x];
  __syncthreads();
  const int64_t pix = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (pix >= npix) return;
  f32x4 raw[2][CM / 4];
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int q = 0; q < CM / 4; ++q) raw[e][q] = *reinterpret_cast<const f32x4*>((e == 0 ? pa : pb) + pix * CM + 4 * q);
  f32x2 total[H2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    float sum = 0.f;
#pragma unroll
    for (int q = 0; q < CM / 4; ++q) {
      sum += raw[e][q].x;
      sum += raw[e][q].y;
      sum += raw[e][q].z;
      sum += raw[e][q].w;
    }
    const float rs = xv_fast_rcp(sum);
    const f32x2 rs2 = f32x2{rs, rs};
    f32x2 lg[H2];
#pragma unroll
    for (int j = 0; j < H2; ++j) {  // renormalise, then log(1e-20 + p)
      const f32x4 r = raw[e][j >> 1];
      const f32x2 x2 = (j & 1) ? f32x2{r.z, r.w} : f32x2{r.x, r.y};
      const f32x2 t = __builtin_elementwise_fma(x2, rs2, f32x2{1e-20f, 1e-20f});
      const f32x2 l = f32x2{__builtin_amdgcn_logf(t.x), __builtin_amdgcn_logf(t.y)};
      lg[j] = l * f32x2{0.6931471805599453f, 0.6931471805599453f};
    }
    f32x2 dot[H2];
#pragma unroll
    for (int cp = 0; cp < H2; ++cp) dot[cp] = f32x2{0.f, 0.f};
#pragma unroll
    for (int k = 0; k < CM; ++k) {
      const f32x2* rowk = reinterpret_cast<const f32x2*>(tab + (e * CM + k) * CM);
      const float lk = lg[k >> 1][k & 1];
#pragma unroll
      for (int cp = 0; cp < H2; ++cp) dot[cp] = __builtin_elementwise_fma(rowk[cp], f32x2{lk, lk}, dot[cp]);
    }
    const f32x2* ln2 = reinterpret_cast<const f32x2*>(ln + e * CM);
#pragma unroll
    for (int cp = 0; cp < H2; ++cp) {
      const f32x2 L = dot[cp] - ln2[cp];
      total[cp] = e == 0 ? L : total[cp] + L;
    }
  }
  const f32x2* lp2 = reinterpret_cast<const f32x2*>(lp);
  float best = 0.f;
  int bi = 0;
  f32x2 v[H2];
#pragma unroll
  for (int j = 0; j < H2; ++j) {
    v[j] = total[j] + lp2[j];
    if (j == 0 || v[j].x > best) best = v[j].x, bi = 2 * j;
    if (v[j].y > best) best = v[j].y, bi = 2 * j + 1;
  }
  if (score_out) {
#pragma unroll
    for (int q = 0; q < CM / 4; ++q)
      *reinterpret_cast<f32x4*>(score_out + pix * CM + 4 * q) = f32x4{v[2 * q].x, v[2 * q].y, v[2 * q + 1].x, v[2 * q + 1].y};
  }
  fused[pix] = bi;
}

// average_mix.py:18-21: argmax of the mean of the experts' probabilities
template <int CMAX>
__global__ __launch_bounds__(256) void average_fuse_kernel(ProbPtrs probs, int E, int C, int64_t npix,
                                                          int64_t* __restrict__ fused, int vec) {
  for (int64_t pix = (int64_t)blockIdx.x * 256 + threadIdx.x; pix < npix; pix += (int64_t)gridDim.x * 256) {
    float s[CMAX];
    load_row<CMAX>(probs.p[0] + pix * C, C, vec, s);
    for (int e = 1; e < E; ++e) {
      float x[CMAX];
      load_row<CMAX>(probs.p[e] + pix * C, C, vec, x);
#pragma unroll
      for (int k = 0; k < CMAX; ++k) s[k] = s[k] + x[k];
    }
    float best = 0.f;
    int bi = 0;
#pragma unroll
    for (int k = 0; k < CMAX; ++k) {
      const float v = s[k] / (float)E;
      if (k < C && (k == 0 || v > best)) {
        best = v;
        bi = k;
      }
    }
    fused[pix] = bi;
  }
}

// dirichlet_mix.py:142-163: S[label][k] += log(1e-10 + p[k]); counts[label] += 1.
// A workgroup keeps `rep` double copies of the [C][C] table (and of the counts) in LDS (16 for C <= 20, fewer beyond: the
// host fits them into 64 KB), copy = lane & (rep - 1) in the fastest-varying position: the lanes of a wave spread over the
// copies whatever their labels are -- only lanes that share a copy can meet on an address.  (One table per workgroup, as
// before round 5, serialised every wave on the few rows its labels select: 12 same-address double atomics per pixel, 0.31
// of HBM.)  fp32 logs added in double everywhere -- LDS copies, the workgroup's sum, the global atomics -- so the totals do
// not depend on how the pixels fall onto workgroups beyond fp64 rounding (tests/test_config4_gpu.py: two ranks = one
// process to 1e-9).  One global double atomic per cell and workgroup.
// copies of a [cells] table of `bytes`-sized elements that fit 64 KB of LDS (two workgroups per CU), a power of two <= 32
inline int table_copies(size_t cells, size_t bytes) {
  int rep = bytes == 8 ? 16 : 32;
  while (rep > 1 && cells * rep * bytes > 64 * 1024) rep >>= 1;
  return rep;
}

__global__ __launch_bounds__(1024) void suffstats_kernel(const float* __restrict__ prob, const int32_t* __restrict__ labels,
                                                       int C, int64_t npix, double* __restrict__ S,
                                                       unsigned long long* __restrict__ counts, int XV_REP, int vec) {
  extern __shared__ __attribute__((aligned(16))) double part[];  // [C * C][XV_REP] sums, then [C][XV_REP] counts (u64)
  unsigned long long* cnt = reinterpret_cast<unsigned long long*>(part + C * C * XV_REP);
  const int T = blockDim.x;
  for (int i = threadIdx.x; i < (C * C + C) * XV_REP; i += T) part[i] = 0.0;      // (0.0 and 0ull share their bits)
  __syncthreads();
  const int rep = threadIdx.x & (XV_REP - 1);
  // log(1e-10 + p) on the transcendental unit (xv_fast_log: v_log_f32, ~1 ulp; the argument is never denormal): the
  // library logf cost 15 instructions a value, 180 a pixel -- as much time as the pixel's 52 bytes take to arrive
  for (int64_t pix = (int64_t)blockIdx.x * T + threadIdx.x; pix < npix; pix += (int64_t)gridDim.x * T) {
    const int l = labels[pix];
    const float* row = prob + pix * C;
    if (vec) {
      float x[32];
      load_row<32>(row, C, true, x);           // requested for every pixel: no divergent address arithmetic in front
      if (l >= 0 && l < C) {
        atomicAdd(&cnt[l * XV_REP + rep], 1ull);
        double* dst = part + l * C * XV_REP + rep;
#pragma unroll
        for (int k = 0; k < 32; ++k)
          if (k < C) atomicAdd(dst + k * XV_REP, (double)xv_fast_log(1e-10f + x[k]));
      }
    } else if (l >= 0 && l < C) {
      atomicAdd(&cnt[l * XV_REP + rep], 1ull);
      for (int k = 0; k < C; ++k) atomicAdd(&part[(l * C + k) * XV_REP + rep], (double)xv_fast_log(1e-10f + row[k]));
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < C * C + C; i += T) {
    if (i < C * C) {
      double a = 0.0;
      for (int r = 0; r < XV_REP; ++r) a += part[i * XV_REP + r];
      if (a != 0.0) atomicAdd(&S[i], a);
    } else {
      unsigned long long a = 0ull;
      for (int r = 0; r < XV_REP; ++r) a += cnt[(i - C * C) * XV_REP + r];
      if (a) atomicAdd(&counts[i - C * C], a);
    }
  }
}

// base_model.py:136-151: cm[label][pred] += 1, negative labels dropped.  The same replicated table (u32 copies of [C][C],
// copy = lane & (rep - 1) fastest): one conflict-free ds_add_u32 per pixel; four pixels per thread and step through 16-byte
// loads (one label quad, two prediction pairs) where both maps are 16-byte aligned.
__global__ __launch_bounds__(1024) void confusion_kernel(const int32_t* __restrict__ labels, const int64_t* __restrict__ pred,
                                                        int C, int64_t npix, unsigned long long* __restrict__ cm, int XV_REP,
                                                        int vec) {
  extern __shared__ unsigned int hist[];       // [C * C][XV_REP]
  const int T = blockDim.x;
  for (int i = threadIdx.x; i < C * C * XV_REP; i += T) hist[i] = 0u;
  __syncthreads();
  const int rep = threadIdx.x & (XV_REP - 1);
  auto count = [&](int l, int64_t p) {
    if (l >= 0 && l < C && p >= 0 && p < C) atomicAdd(&hist[(l * C + (int)p) * XV_REP + rep], 1u);
  };
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  typedef long long i64x2 __attribute__((ext_vector_type(2)));
  const int64_t quads = vec ? npix >> 2 : 0, stride = (int64_t)gridDim.x * T;
  // two quads per thread and step: six 16-byte requests in flight per lane (one workgroup of 16 waves per CU: 96 KB)
  for (int64_t q = (int64_t)blockIdx.x * T + threadIdx.x; q < quads; q += 2 * stride) {
    const int64_t q1 = q + stride;
    const bool two = q1 < quads;
    const i32x4 l = *reinterpret_cast<const i32x4*>(labels + 4 * q);
    const i64x2 p0 = *reinterpret_cast<const i64x2*>(pred + 4 * q), p1 = *reinterpret_cast<const i64x2*>(pred + 4 * q + 2);
    i32x4 m = {-1, -1, -1, -1};
    i64x2 r0 = {0, 0}, r1 = {0, 0};
    if (two) {
      m = *reinterpret_cast<const i32x4*>(labels + 4 * q1);
      r0 = *reinterpret_cast<const i64x2*>(pred + 4 * q1), r1 = *reinterpret_cast<const i64x2*>(pred + 4 * q1 + 2);
    }
    count(l.x, p0.x);
    count(l.y, p0.y);
    count(l.z, p1.x);
    count(l.w, p1.y);
    count(m.x, r0.x);
    count(m.y, r0.y);
    count(m.z, r1.x);
    count(m.w, r1.y);
  }
  // the tail of the vector form / the whole map where a pointer is not 16-byte aligned
  for (int64_t pix = 4 * quads + (int64_t)blockIdx.x * T + threadIdx.x; pix < npix; pix += stride) count(labels[pix], pred[pix]);
  __syncthreads();
  for (int i = threadIdx.x; i < C * C; i += T) {
    unsigned long long a = 0ull;
    for (int r = 0; r < XV_REP; ++r) a += hist[i * XV_REP + r];
    if (a) atomicAdd(&cm[i], a);
  }
}

// int64 label map -> one byte per pixel for the trip to the host (predict()'s return value is np.int64 [N,H,W],
// base_model.py:279-288: the host widens it again while it fills the result array).  8 labels per thread: four 16-byte
// reads, one 8-byte store.
__global__ __launch_bounds__(256) void narrow_labels_kernel(const int64_t* __restrict__ in, uint8_t* __restrict__ out, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * 256 * 8;
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8; i < n; i += stride) {
    if (i + 8 <= n) {
      uint64_t packed = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(in + i + 2 * j);     // two labels: low words .x and .z
        packed |= (uint64_t)(v.x & 0xffu) << (16 * j) | (uint64_t)(v.z & 0xffu) << (16 * j + 8);
      }
      *reinterpret_cast<uint64_t*>(out + i) = packed;
    } else {
      for (int64_t k = i; k < n; ++k) out[k] = (uint8_t)in[k];
    }
  }
}

}  // namespace

extern "C" int xv_narrow_labels(const int64_t* labels, int64_t n, uint8_t* out, void* stream) {
  XV_CHECK_ARG(labels && out && ((uintptr_t)labels & 15) == 0 && ((uintptr_t)out & 7) == 0);
  XV_CHECK_SHAPE(n > 0);
  hipLaunchKernelGGL(narrow_labels_kernel, dim3(grid_for((n + 7) / 8, 256, 2048)), dim3(256), 0, (hipStream_t)stream, labels, out, n);
  return xv_launch_status();
}

extern "C" int xv_bayes_fuse(const int64_t* const* labels, int num_experts, const float* loglik, const float* logprior,
                             int num_classes, int64_t npix, int64_t* fused, float* score_out, void* stream) {
  XV_CHECK_ARG(labels && loglik && logprior && fused);
  XV_CHECK_SHAPE(num_experts >= 1 && num_experts <= MAXE && num_classes >= 1 && num_classes <= 32 && npix > 0);
  LabelPtrs lp{};
  for (int e = 0; e < num_experts; ++e) {
    XV_CHECK_ARG(labels[e]);
    lp.p[e] = labels[e];
  }
  hipStream_t s = (hipStream_t)stream;
  const int cm = num_classes <= 16 ? 16 : 32;
  const size_t lds = (size_t)(num_experts * num_classes * cm + cm) * 4;
  if (num_experts == 2 && score_out == nullptr && (((uintptr_t)labels[0] | (uintptr_t)labels[1] | (uintptr_t)fused) & 15) == 0) {
    hipLaunchKernelGGL(bayes_fuse2_kernel<32>, dim3(grid_for((npix + 1) / 2, 256, xv_num_cus() * 8)), dim3(256),
                       (size_t)(3 * num_classes * num_classes + num_classes) * 4, s, labels[0], labels[1], loglik, logprior,
                       num_classes, npix, fused);
    return xv_launch_status();
  }
  if (cm == 16)
    hipLaunchKernelGGL(bayes_fuse_kernel<16>, dim3(grid_for(npix)), dim3(256), lds, s, lp, num_experts, loglik, logprior,
                       num_classes, npix, fused, score_out);
  else
    hipLaunchKernelGGL(bayes_fuse_kernel<32>, dim3(grid_for(npix)), dim3(256), lds, s, lp, num_experts, loglik, logprior,
                       num_classes, npix, fused, score_out);
  return xv_launch_status();
}

extern "C" int xv_bayes_fuse_lut(const int64_t* label_a, const int64_t* label_b, const int64_t* lut, int num_classes,
                                 int64_t npix, int64_t* fused, void* stream) {
  XV_CHECK_ARG(label_a && label_b && lut && fused);
  XV_CHECK_SHAPE(num_classes >= 1 && num_classes <= 64 && npix > 0);
  hipLaunchKernelGGL(bayes_lut_kernel, dim3(grid_for(npix)), dim3(256), (size_t)num_classes * num_classes * 4,
                     (hipStream_t)stream, label_a, label_b, lut, num_classes, npix, fused);
  return xv_launch_status();
}

extern "C" int xv_dirichlet_fuse(const float* const* probs, int num_experts, const float* am1, const float* lognorm,
                                 const float* logprior, int num_classes, int64_t npix, int64_t* fused, float* score_out,
                                 void* stream) {
  XV_CHECK_ARG(probs && am1 && lognorm && logprior && fused);
  XV_CHECK_SHAPE(num_experts >= 1 && num_experts <= MAXE && num_classes >= 1 && num_classes <= 32 && npix > 0);
  ProbPtrs pp{};
  for (int e = 0; e < num_experts; ++e) {
    XV_CHECK_ARG(probs[e]);
    pp.p[e] = probs[e];
  }
  hipStream_t s = (hipStream_t)stream;
  const int cm = num_classes <= 16 ? 16 : 32;
  const size_t lds = (size_t)(num_experts * num_classes * cm + num_experts * cm + cm) * 4;
  int vec = (num_classes & 3) == 0;
  for (int e = 0; e < num_experts; ++e) vec = vec && ((uintptr_t)probs[e] & 15) == 0;
  // XV_DIRICHLET_FUSE_PK=0: the scalar form for the two-expert 12-class case too (read per call: the test that pins the two
  // forms to the same bits switches it)
  const char* pk_env = getenv("XV_DIRICHLET_FUSE_PK");
  const bool pk = !(pk_env && pk_env[0] == '0');
  const bool score_ok = !score_out || ((uintptr_t)score_out & 15) == 0;
  if (num_classes == 12 && vec && num_experts == 2 && score_ok && pk)
    hipLaunchKernelGGL(dirichlet_fuse_pk_kernel<12>, dim3((unsigned)((npix + 255) / 256)), dim3(256),
                       (size_t)(2 * 12 * 12 + 2 * 12 + 12) * 4, s, pp.p[0], pp.p[1], am1, lognorm, logprior, npix, fused,
                       score_out);
  else if (num_classes == 12 && vec)
    hipLaunchKernelGGL((dirichlet_fuse_kernel<12, true>), dim3(grid_for(npix, 256, xv_num_cus() * 8)), dim3(256),
                       (size_t)(num_experts * 12 * 12 + num_experts * 12 + 12) * 4, s, pp, num_experts, am1, lognorm, logprior,
                       num_classes, npix, fused, score_out, vec);
  else if (cm == 16)
    hipLaunchKernelGGL(dirichlet_fuse_kernel<16>, dim3(grid_for(npix)), dim3(256), lds, s, pp, num_experts, am1, lognorm,
                       logprior, num_classes, npix, fused, score_out, vec);
  else
    hipLaunchKernelGGL(dirichlet_fuse_kernel<32>, dim3(grid_for(npix)), dim3(256), lds, s, pp, num_experts, am1, lognorm,
                       logprior, num_classes, npix, fused, score_out, vec);
  return xv_launch_status();
}

extern "C" int xv_average_fuse(const float* const* probs, int num_experts, int num_classes, int64_t npix,
                               int64_t* fused, void* stream) {
  XV_CHECK_ARG(probs && fused);
  XV_CHECK_SHAPE(num_experts >= 1 && num_experts <= MAXE && num_classes >= 1 && num_classes <= 32 && npix > 0);
  ProbPtrs pp{};
  for (int e = 0; e < num_experts; ++e) {
    XV_CHECK_ARG(probs[e]);
    pp.p[e] = probs[e];
  }
  hipStream_t s = (hipStream_t)stream;
  int vec = (num_classes & 3) == 0;
  for (int e = 0; e < num_experts; ++e) vec = vec && ((uintptr_t)probs[e] & 15) == 0;
  if (num_classes <= 16)
    hipLaunchKernelGGL(average_fuse_kernel<16>, dim3(grid_for(npix)), dim3(256), 0, s, pp, num_experts, num_classes, npix,
                       fused, vec);
  else
    hipLaunchKernelGGL(average_fuse_kernel<32>, dim3(grid_for(npix)), dim3(256), 0, s, pp, num_experts, num_classes, npix,
                       fused, vec);
  return xv_launch_status();
}

extern "C" int xv_dirichlet_suffstats(const float* prob, const int32_t* labels, int num_classes, int64_t npix, double* S,
                                      int64_t* counts, void* stream) {
  XV_CHECK_ARG(prob && labels && S && counts);
  XV_CHECK_SHAPE(num_classes >= 1 && num_classes <= 64 && npix > 0);
  const size_t cells = (size_t)num_classes * num_classes + num_classes;
  const int rep = table_copies(cells, 8);
  const size_t lds = cells * rep * 8;                        // 20 KB at C = 12
  const int vec = (num_classes & 3) == 0 && num_classes <= 32 && ((uintptr_t)prob & 15) == 0;
  // few workgroups: each ends with C*C same-address double atomics, which serialise across the chip (~12 ns each)
  // one 16-wave workgroup per CU
  hipLaunchKernelGGL(suffstats_kernel, dim3(grid_for(npix, 1024, xv_num_cus())), dim3(1024), lds, (hipStream_t)stream,
                     prob, labels, num_classes, npix, S, reinterpret_cast<unsigned long long*>(counts), rep, vec);
  return xv_launch_status();
}

extern "C" int xv_confusion_matrix(const int32_t* labels, const int64_t* pred, int num_classes, int64_t npix,
                                   int64_t* cm, void* stream) {
  XV_CHECK_ARG(labels && pred && cm);
  XV_CHECK_SHAPE(num_classes >= 1 && num_classes <= 64 && npix > 0);
  const int rep = table_copies((size_t)num_classes * num_classes, 4);
  const size_t lds = (size_t)num_classes * num_classes * rep * 4;        // 18 KB at C = 12
  const int vec = ((uintptr_t)labels & 15) == 0 && ((uintptr_t)pred & 15) == 0;
  // one 16-wave workgroup per CU: each ends with C*C same-address atomics, which serialise across the chip (~12 ns each)
  hipLaunchKernelGGL(confusion_kernel, dim3(grid_for((npix + 7) / 8, 1024, xv_num_cus())), dim3(1024), lds,
                     (hipStream_t)stream, labels, pred, num_classes, npix, reinterpret_cast<unsigned long long*>(cm), rep, vec);
  return xv_launch_status();
}
