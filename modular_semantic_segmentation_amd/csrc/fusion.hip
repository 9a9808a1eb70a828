// Per-pixel probabilistic fusion, sufficient statistics and confusion matrix for gfx950.
// All HBM-bound: one thread per pixel, class vectors in registers, tables in LDS.
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
template <int CMAX>
__global__ __launch_bounds__(256) void dirichlet_fuse_kernel(ProbPtrs probs, int E, const float* __restrict__ am1,
                                                            const float* __restrict__ lognorm,
                                                            const float* __restrict__ logprior, int C, int64_t npix,
                                                            int64_t* __restrict__ fused, float* __restrict__ score_out) {
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
    for (int e = 0; e < E; ++e) {
      float lx[CMAX];
      float sum = 0.f;
#pragma unroll
      for (int k = 0; k < CMAX; ++k) {
        lx[k] = k < C ? probs.p[e][pix * C + k] : 0.f;
        sum += lx[k];
      }
      const float rs = xv_fast_rcp(sum);  // (the same helper forms as the fused head: xv_common.h)
#pragma unroll
      for (int k = 0; k < CMAX; ++k) lx[k] = k < C ? xv_fast_log(1e-20f + lx[k] * rs) : 0.f;  // renormalise, then log(1e-20 + p)
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

// average_mix.py:18-21: argmax of the mean of the experts' probabilities
template <int CMAX>
__global__ __launch_bounds__(256) void average_fuse_kernel(ProbPtrs probs, int E, int C, int64_t npix,
                                                          int64_t* __restrict__ fused) {
  for (int64_t pix = (int64_t)blockIdx.x * 256 + threadIdx.x; pix < npix; pix += (int64_t)gridDim.x * 256) {
    float s[CMAX];
#pragma unroll
    for (int k = 0; k < CMAX; ++k) s[k] = 0.f;
    for (int e = 0; e < E; ++e)
#pragma unroll
      for (int k = 0; k < CMAX; ++k)
        if (k < C) s[k] = e == 0 ? probs.p[e][pix * C + k] : s[k] + probs.p[e][pix * C + k];
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
// Per-block float64 partials in LDS (ds_add_f64), one global f64 atomic per cell per block.
__global__ __launch_bounds__(256) void suffstats_kernel(const float* __restrict__ prob, const int32_t* __restrict__ labels,
                                                       int C, int64_t npix, double* __restrict__ S,
                                                       unsigned long long* __restrict__ counts) {
  extern __shared__ __attribute__((aligned(16))) double part[];  // [C][C] then counts [C] (as double bits of u64)
  unsigned long long* cnt = reinterpret_cast<unsigned long long*>(part + C * C);
  for (int i = threadIdx.x; i < C * C; i += 256) part[i] = 0.0;
  for (int i = threadIdx.x; i < C; i += 256) cnt[i] = 0ull;
  __syncthreads();
  for (int64_t pix = (int64_t)blockIdx.x * 256 + threadIdx.x; pix < npix; pix += (int64_t)gridDim.x * 256) {
    const int l = labels[pix];
    if (l >= 0 && l < C) {
      atomicAdd(&cnt[l], 1ull);
      for (int k = 0; k < C; ++k) atomicAdd(&part[l * C + k], (double)logf(1e-10f + prob[pix * C + k]));
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < C * C; i += 256)
    if (part[i] != 0.0) atomicAdd(&S[i], part[i]);
  for (int i = threadIdx.x; i < C; i += 256)
    if (cnt[i]) atomicAdd(&counts[i], cnt[i]);
}

// base_model.py:136-151: cm[label][pred] += 1, negative labels dropped.
__global__ __launch_bounds__(256) void confusion_kernel(const int32_t* __restrict__ labels, const int64_t* __restrict__ pred,
                                                       int C, int64_t npix, unsigned long long* __restrict__ cm) {
  extern __shared__ unsigned int hist[];
  for (int i = threadIdx.x; i < C * C; i += 256) hist[i] = 0u;
  __syncthreads();
  for (int64_t pix = (int64_t)blockIdx.x * 256 + threadIdx.x; pix < npix; pix += (int64_t)gridDim.x * 256) {
    const int l = labels[pix];
    const int64_t p = pred[pix];
    if (l >= 0 && l < C && p >= 0 && p < C) atomicAdd(&hist[l * C + (int)p], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < C * C; i += 256)
    if (hist[i]) atomicAdd(&cm[i], (unsigned long long)hist[i]);
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
  if (cm == 16)
    hipLaunchKernelGGL(dirichlet_fuse_kernel<16>, dim3(grid_for(npix)), dim3(256), lds, s, pp, num_experts, am1, lognorm,
                       logprior, num_classes, npix, fused, score_out);
  else
    hipLaunchKernelGGL(dirichlet_fuse_kernel<32>, dim3(grid_for(npix)), dim3(256), lds, s, pp, num_experts, am1, lognorm,
                       logprior, num_classes, npix, fused, score_out);
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
  if (num_classes <= 16)
    hipLaunchKernelGGL(average_fuse_kernel<16>, dim3(grid_for(npix)), dim3(256), 0, s, pp, num_experts, num_classes, npix,
                       fused);
  else
    hipLaunchKernelGGL(average_fuse_kernel<32>, dim3(grid_for(npix)), dim3(256), 0, s, pp, num_experts, num_classes, npix,
                       fused);
  return xv_launch_status();
}

extern "C" int xv_dirichlet_suffstats(const float* prob, const int32_t* labels, int num_classes, int64_t npix, double* S,
                                      int64_t* counts, void* stream) {
  XV_CHECK_ARG(prob && labels && S && counts);
  XV_CHECK_SHAPE(num_classes >= 1 && num_classes <= 64 && npix > 0);
  const size_t lds = (size_t)(num_classes * num_classes + num_classes) * 8;
  hipLaunchKernelGGL(suffstats_kernel, dim3(grid_for(npix, 256, 1024)), dim3(256), lds, (hipStream_t)stream, prob, labels,
                     num_classes, npix, S, reinterpret_cast<unsigned long long*>(counts));
  return xv_launch_status();
}

extern "C" int xv_confusion_matrix(const int32_t* labels, const int64_t* pred, int num_classes, int64_t npix,
                                   int64_t* cm, void* stream) {
  XV_CHECK_ARG(labels && pred && cm);
  XV_CHECK_SHAPE(num_classes >= 1 && num_classes <= 64 && npix > 0);
  hipLaunchKernelGGL(confusion_kernel, dim3(grid_for(npix, 256, 1024)), dim3(256),
                     (size_t)num_classes * num_classes * 4, (hipStream_t)stream, labels, pred, num_classes, npix,
                     reinterpret_cast<unsigned long long*>(cm));
  return xv_launch_status();
}
