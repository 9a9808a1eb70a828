// Shared device/host helpers for libxview_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <initializer_list>

#include "../../include/xview_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define XV_CHECK_ARG(cond) \
  do {                     \
    if (!(cond)) return XV_EINVAL; \
  } while (0)
#define XV_CHECK_SHAPE(cond) \
  do {                       \
    if (!(cond)) return XV_ESHAPE; \
  } while (0)

// bf16-only entry points (everything except the forward convolutions, which read / write e4m3 maps too): an XV_FP8
// descriptor handed to them would be reinterpreted as bf16 -- garbage and reads past the end of a map of half the bytes.
// Null descriptors (optional arguments) pass; the entry point's own argument checks deal with them.
static inline bool xv_all_bf16(std::initializer_list<const xv_act*> acts) {
  for (const xv_act* a : acts)
    if (a != nullptr && a->dtype != XV_BF16) return false;
  return true;
}
#define XV_REQUIRE_BF16(...)                               \
  do {                                                     \
    if (!xv_all_bf16({__VA_ARGS__})) return XV_EINVAL;     \
  } while (0)

static inline int xv_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? XV_OK : (int)e;
}

// Per-device caches (one process normally drives one GPU, but nothing here may depend on it): up to XV_MAX_DEVICES devices
// of one process; an out-of-range device index falls back to querying every time.
#define XV_MAX_DEVICES 16
static inline int xv_current_device() {
  int dev = 0;
  return hipGetDevice(&dev) == hipSuccess ? dev : 0;
}

// CU count of the current device (cached per device; 256 on MI355X) -- sizes persistent grids.
static inline int xv_num_cus() {
  static int cus[XV_MAX_DEVICES] = {0};
  const int dev = xv_current_device();
  if (dev >= 0 && dev < XV_MAX_DEVICES && cus[dev] > 0) return cus[dev];
  int n = 0;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
  if (dev >= 0 && dev < XV_MAX_DEVICES) cus[dev] = n;
  return n;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per device and per kernel: `done` is the kernel's own flag array
static inline hipError_t xv_allow_dynamic_lds(const void* kernel, int bytes, bool (&done)[XV_MAX_DEVICES],
                                              bool from_offset_0 = true) {
  const int dev = xv_current_device();
  if (dev >= 0 && dev < XV_MAX_DEVICES && done[dev]) return hipSuccess;
  hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess && from_offset_0) {
    // The conv kernels behind this helper address their dynamic LDS from offset 0 (LDS-DMA destinations in M0, ds_read offsets in
    // assembly).  A static LDS object in front of it -- the compiler promotes an alloca into LDS when a struct does not stay
    // in registers: round 4, a bool member in the tile record -- would sit on top of the first operand buffer: refuse to launch.
    hipFuncAttributes attr;
    e = hipFuncGetAttributes(&attr, kernel);
    if (e == hipSuccess && attr.sharedSizeBytes != 0) e = hipErrorInvalidConfiguration;
  }
  if (e == hipSuccess && dev >= 0 && dev < XV_MAX_DEVICES) done[dev] = true;
  return e;
}

// A map this library can address at all: positive dimensions, sides below 2^24, fewer than 2^31 padded rows and 2^36
// padded pixels (64 G pixels: beyond any 288 GB buffer).  The size calculators return 0 and the choosers XV_ESHAPE past it, so that their
// 64-bit arithmetic cannot overflow (tools/asan_host_check.py fuzzes them under UBSan).
static inline bool xv_dims_sane(int n, int h, int w) {
  if (n <= 0 || h <= 0 || w <= 0 || h >= (1 << 24) || w >= (1 << 24)) return false;
  const int64_t rows = (int64_t)n * ((int64_t)h + 2);
  return rows < ((int64_t)1 << 31) && rows * ((int64_t)w + 2) < ((int64_t)1 << 36);
}

// padded-NHWC geometry helpers
__host__ __device__ static inline int64_t xv_row_pitch(int w, int c) { return (int64_t)(w + 2) * c; }
__host__ __device__ static inline int64_t xv_img_pitch(int h, int w, int c) {
  return (int64_t)(h + 2) * (w + 2) * c;
}

// 16-byte-slot swizzle shared by the weight packer and the conv kernels: slot s of the 128-byte
// LDS row of weight row / patch column `row` lives at slot s ^ (row & 6).  Two consecutive 128-B
// rows fill one 256-B LDS bank row; a ds_read_b128 is served in 16-lane groups that, for the
// 16x16x32 MFMA fragment map, hold 16 consecutive rows with two adjacent logical slots
// ({0-3,12-15} slot a, {4-11} slot a+1).  An exhaustive search over GF(2)-linear maps of the row
// index (tools/lds_swizzle_search.py) shows `row & 6` is conflict-free for EVERY starting row, i.e.
// for all three horizontal taps of a 3x3 conv; the earlier (row>>1)&7 was 2-way for 3 of 4 starts.
// For activation patches `row` is the pixel's COLUMN inside the patch (the patch pitch is even, so
// bank-row parity is preserved), which makes the swizzle independent of the patch row and lets
// every fragment read be base-register + immediate.
__host__ __device__ static inline int xv_swz(int row, int slot) { return slot ^ (row & 6); }

// 64-byte rows (generation-2 conv kernel): slot s of row `row` lives at s ^ ((row >> 1) & 2)
__host__ __device__ static inline int xv_swz32(int row, int slot) { return slot ^ ((row >> 1) & 2); }

// four fp32 -> four OCP e4m3 bytes (round-to-nearest-even) of value * mul, saturating at the largest finite e4m3 (the
// conversion's own overflow behaviour depends on a mode bit, so the clamp is explicit)
__device__ static inline uint32_t xv_pack_fp8x4(float v0, float v1, float v2, float v3, float mul) {
  const float a = __builtin_amdgcn_fmed3f(v0 * mul, -448.f, 448.f);
  const float b = __builtin_amdgcn_fmed3f(v1 * mul, -448.f, 448.f);
  const float c = __builtin_amdgcn_fmed3f(v2 * mul, -448.f, 448.f);
  const float d = __builtin_amdgcn_fmed3f(v3 * mul, -448.f, 448.f);
  int p = 0;
  p = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, p, false);
  p = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, p, true);
  return (uint32_t)p;
}

__device__ static inline uint32_t pack_bf16x2(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
  bf16x2 v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(uint32_t, v);
}

__device__ static inline float bf16_bits_to_f32(uint32_t bits16) {
  return __builtin_bit_cast(float, bits16 << 16);
}

// Output-side helpers shared by the MFMA conv kernels (a lane holds 4 consecutive output channels of one pixel for
// each of the four 16-channel blocks).
__device__ static __forceinline__ void xv_pair16(const u32x2 a, const u32x2 b, u32x4& out) {
  // lanes of row r (= lane >> 4) hold channel groups r*4 of `a` (block j) and `b` (block j+1); after the swap a
  // lane holds 8 consecutive channels: rows 0..3 -> channel offsets 0, 16, 8, 24 of the 32-channel pair
  const auto s0 = __builtin_amdgcn_permlane16_swap(a.x, b.x, false, false);
  const auto s1 = __builtin_amdgcn_permlane16_swap(a.y, b.y, false, false);
  out = u32x4{s0[0], s1[0], s0[1], s1[1]};
}

// relu and 2x2 max on PACKED bf16 pairs: a non-negative bf16 orders like its bit pattern as a signed 16-bit
// integer and every negative one (sign bit) is a negative integer, so relu is v_pk_max_i16 against 0 and, after
// it, so is the max-pool -- two values per instruction and none of the NaN-quieting v_max_f32 x,x,x that fmaxf
// costs under IEEE mode.  Rounding to bf16 is monotone, so pooling after rounding equals rounding after pooling.
__device__ static __forceinline__ uint32_t pk_max_i16(uint32_t a, uint32_t b) {
  typedef short s16x2 __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b)));
}

// The per-pixel softmax / log-probability arithmetic of the heads (tf.nn.softmax, Dirichlet.log_prob: basic_fusion_model.py:
// 21-22, dirichlet_mix.py:23-36) on the transcendental unit: v_exp_f32 / v_log_f32 / v_rcp_f32 (about 1 ulp each, denormal
// inputs / results flushed) in place of the correctly rounded library forms (10-15 instructions each; an IEEE division is ten).
// xv_fast_exp(x) = exp2(x * log2 e): the rounding of the product is amplified by |x|, so its relative error GROWS as
// ~|x| * 6e-8 (6e-7 at x = -10, 5e-6 at x = -80) -- terms that small are below fp32 resolution of the softmax sum they
// enter.  Probabilities, Dirichlet log-probabilities and the training loss are therefore NOT correctly rounded libm
// results (numerics notes); the accuracy tests bound what that costs.  24 logs, 24 exps and 48 divisions
// per pixel made the fused Dirichlet head 10x slower than its 288 FMAs.  Every kernel that must agree
// bit for bit with another one (fused head <-> head + fusion kernels) uses the same helper.
__device__ static __forceinline__ float xv_fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ static __forceinline__ float xv_fast_log(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }
__device__ static __forceinline__ float xv_fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// Order-preserving map of packed bf16 bit patterns onto signed 16-bit integers (an involution: negative values have their
// magnitude bits flipped), for a 2x2 max on packed pairs WITHOUT a preceding relu.
__device__ static __forceinline__ uint32_t pk_ord_bf16(uint32_t x) {
  typedef short s16x2 __attribute__((ext_vector_type(2)));
  const s16x2 v = __builtin_bit_cast(s16x2, x);
  const s16x2 m = (v >> 15) & (short)0x7fff;
  return __builtin_bit_cast(uint32_t, v ^ m);
}

// value of lane ^ 1 (DPP quad_perm [1,0,3,2]) of a packed register
__device__ static __forceinline__ uint32_t pk_dpp_swap1(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xf, 0xf, true);
}

// ---- diagnostic build only (-DXV_CLOCK_STAMP: `make stamp` -> tools/build/libxview_hip_stamp.so, read by tools/conv_clock.py;
// never in the shipped library) --------------------------------------------------------------------------------------------
// The clock the chip holds inside a conv kernel's item loop = delta s_memtime (shader cycles) / delta s_memrealtime (a
// constant 100 MHz counter) x 100 MHz, stamped ONCE around the loop by every workgroup (MI355X_MICROARCH.md, 'DVFS
// give-back' item 6).  The four values go to a buffer of their own after the loop; nothing is computed from them.
#ifdef XV_CLOCK_STAMP
#define XV_CLK_SLOTS 2048
#define XV_CLK_BEGIN()                                              \
  const unsigned long long clk_m0 = __builtin_amdgcn_s_memtime();   \
  const unsigned long long clk_r0 = __builtin_amdgcn_s_memrealtime();
#define XV_CLK_END(buf)                                                                   \
  {                                                                                       \
    const unsigned long long clk_m1 = __builtin_amdgcn_s_memtime();                       \
    const unsigned long long clk_r1 = __builtin_amdgcn_s_memrealtime();                   \
    if (threadIdx.x == 0 && blockIdx.x < XV_CLK_SLOTS) {                                  \
      buf[blockIdx.x * 4 + 0] = clk_m0, buf[blockIdx.x * 4 + 1] = clk_r0;                 \
      buf[blockIdx.x * 4 + 2] = clk_m1, buf[blockIdx.x * 4 + 3] = clk_r1;                 \
    }                                                                                     \
  }
#else
#define XV_CLK_BEGIN()
#define XV_CLK_END(buf)
#endif
