// 1x1 convolution as a flat GEMM for gfx950: Y[m][co] = act(sum_ci X[m][ci] W[ci][co] + b[co]) (+ addend, mask).
//
// The ResNet-style AdapNet expert (adapnet.py:12-173) is mostly 1x1 convs over 256..9216 input channels.  A padded
// NHWC activation is a dense row-major [Mp = N (H+2) (W+2)][C] matrix, so the conv is a plain GEMM over ALL padded
// pixels with the stores predicated to the interior (the zero border is never written; 3-13 % extra rows).  No 2-D
// tiling, no halo: a workgroup owns 128 consecutive pixel rows x 128 output channels.
//
//   * 4 waves (2 x 2), each 64 pixels x 64 channels = 4 x 4 tiles of v_mfma_f32_16x16x32_bf16 (weights as the A
//     operand: a lane ends up with 4 consecutive output channels of one pixel, an 8-byte NHWC store);
//   * K in 64-channel steps: both operand tiles (128 rows x 128 B each) are copied global -> LDS by LDS-DMA
//     (`global_load_lds_dwordx4`, 1 KB per instruction, 8 per wave and step), double-buffered: step t+1 streams in
//     while step t's 32 MFMAs per wave run; ONE barrier per step;
//   * 64 KB LDS and <= 128 VGPRs: two workgroups per CU, so one workgroup's barrier / epilogue hides under the
//     other's matrix work;
//   * LDS rows are 128 B (32 banks): row r keeps 16-byte slot s at s ^ ((r >> 1) & 7), applied on the GLOBAL side of
//     the DMA (the copy itself is linear), which makes every 16-lane phase of a ds_read_b128 fragment read hit all
//     64 banks once;
//   * consecutive workgroups of one XCD share a pixel tile (blockIdx is remapped so that an XCD's L2 sees each
//     activation row block once per output-channel sweep).
// Accumulation order per output = ascending 32-channel blocks, the same as the other conv kernels: bit-identical.
#include "xv_common.h"

namespace {

struct GemmArgs {
  const __bf16* x;
  const __bf16* wpk;  // packed image 1: [cin/64][cout][8 slots][8], slot s of row co stored at s ^ (co & 6)
  const float* bias;
  __bf16* y;
  const __bf16* mask;
  const __bf16* addend;
  int64_t Mp;
  int H, W, Cin, Cout;
  int relu;
  int n_tiles;
  int nblk;
};

constexpr int G_BM = 128, G_BN = 128;
constexpr int G_TILE_BYTES = 128 * 128;          // one operand tile of one K step
constexpr int G_STAGE_BYTES = 2 * G_TILE_BYTES;  // weights, then pixels
constexpr int G_LDS_BYTES = 2 * G_STAGE_BYTES;

__global__ __launch_bounds__(256, 2) void conv1x1_gemm_kernel(GemmArgs a) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 1, wc = wave & 1;

  // workgroups are dealt round-robin to the 8 XCDs: give each XCD a contiguous run of the (pixel tile, channel
  // tile) list, channel tiles fastest
  int bid = blockIdx.x;
  if ((a.nblk & 7) == 0) bid = (bid & 7) * (a.nblk >> 3) + (bid >> 3);
  const int mt = bid / a.n_tiles, nt = bid - mt * a.n_tiles;
  const int64_t m0 = (int64_t)mt * G_BM;
  const int n0 = nt * G_BN;
  const int Cin = a.Cin, Cout = a.Cout;
  const int nsteps = Cin >> 6;

  // ---- DMA addressing: piece p of a tile = rows 8p .. 8p+7, lane -> (row, 16-byte slot)
  const int drow = lane >> 3, dslot = lane & 7;
  int woff[4], xoff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (wave * 4 + i) * 8 + drow;
    const int g = (r >> 1) & 7;
    woff[i] = ((n0 + r) << 7) + ((dslot ^ g ^ (r & 6)) << 4);
    int64_t m = m0 + r;
    if (m >= a.Mp) m = a.Mp - 1;  // rows past the end: any valid row, never stored
    xoff[i] = (int)(m - m0) * Cin * 2 + ((dslot ^ g) << 4);
  }
  const char* wbase = reinterpret_cast<const char*>(a.wpk);
  const char* xbase = reinterpret_cast<const char*>(a.x) + m0 * Cin * 2;
  auto issue = [&](int step, int stage) {
    const char* ws = wbase + ((int64_t)step * Cout << 7);
    const char* xs = xbase + (step << 7);
    const int dst = stage * G_STAGE_BYTES + wave * 4096;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst + i * 1024), "v"(woff[i]), "s"(ws)
                   : "memory");
#pragma unroll
    for (int i = 0; i < 4; ++i)
      asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst + G_TILE_BYTES + i * 1024),
                   "v"(xoff[i]), "s"(xs)
                   : "memory");
  };

  // ---- fragment addressing: lane -> row t = lane % 16 of a 16-row tile, K slot q = lane / 16 (+4 for the second
  // 32-channel half), swizzled by the row
  const int t = lane & 15, q = lane >> 4, g = t >> 1;
  const int fo0 = t * 128 + (((q ^ (g & 3)) + ((g >> 2) << 2)) << 4);
  const int fo1 = t * 128 + (((q ^ (g & 3)) + (((g >> 2) ^ 1) << 2)) << 4);
  const int wfrag = wc * 64 * 128, xfrag = G_TILE_BYTES + wr * 64 * 128;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  issue(0, 0);
  for (int step = 0; step < nsteps; ++step) {
    const int stage = step & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // this step's tiles have landed for every wave; the other stage is free again
    if (step + 1 < nsteps) issue(step + 1, stage ^ 1);
    const char* sb = smem + stage * G_STAGE_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int fo = kk ? fo1 : fo0;
      bf16x8 wf[4], xf[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(sb + wfrag + j * 2048 + fo);
#pragma unroll
      for (int i = 0; i < 4; ++i) xf[i] = *reinterpret_cast<const bf16x8*>(sb + xfrag + i * 2048 + fo);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc[i][j], 0, 0, 0);
    }
  }

  // ---- epilogue: bias, activation, addend, mask (the order of the other conv kernels), interior pixels only
  const int Wp = a.W + 2, Hp = a.H + 2;
  const int cb = n0 + wc * 64 + q * 4;
  f32x4 bj[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) bj[j] = *reinterpret_cast<const f32x4*>(a.bias + cb + j * 16);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = (int)m0 + wr * 64 + i * 16 + t;  // Mp < 2^31 (checked by the launcher)
    const int row = m / Wp;
    const int xx = m - row * Wp;
    const int yy = row % Hp;
    if (m >= a.Mp || xx < 1 || xx > a.W || yy < 1 || yy > a.H) continue;
    const int64_t off = (int64_t)m * Cout + cb;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x4 v = acc[i][j] + bj[j];
      if (a.relu) {
        v.x = fmaxf(v.x, 0.f);
        v.y = fmaxf(v.y, 0.f);
        v.z = fmaxf(v.z, 0.f);
        v.w = fmaxf(v.w, 0.f);
      }
      if (a.addend != nullptr) {
        const u32x2 ad = *reinterpret_cast<const u32x2*>(a.addend + off + j * 16);
        v.x += bf16_bits_to_f32(ad.x & 0xffffu);
        v.y += __builtin_bit_cast(float, ad.x & 0xffff0000u);
        v.z += bf16_bits_to_f32(ad.y & 0xffffu);
        v.w += __builtin_bit_cast(float, ad.y & 0xffff0000u);
      }
      if (a.mask != nullptr) {
        const u32x2 mk = *reinterpret_cast<const u32x2*>(a.mask + off + j * 16);
        v.x = bf16_bits_to_f32(mk.x & 0xffffu) > 0.f ? v.x : 0.f;
        v.y = __builtin_bit_cast(float, mk.x & 0xffff0000u) > 0.f ? v.y : 0.f;
        v.z = bf16_bits_to_f32(mk.y & 0xffffu) > 0.f ? v.z : 0.f;
        v.w = __builtin_bit_cast(float, mk.y & 0xffff0000u) > 0.f ? v.w : 0.f;
      }
      *reinterpret_cast<u32x2*>(a.y + off + j * 16) = u32x2{pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w)};
    }
  }
}

// ---- narrow form (tile configuration 23): 64 output channels, 64 pixel rows per workgroup ---------------------------
// The two score convs of the FCN (score_conv4 / score_conv5, simple_fcn.py:69-79: 512 -> num_units = 64 channels on the
// 1/8- and 1/16-resolution maps) are tiny GEMMs: 1.2-4.8 GFLOP on 18 k-74 k pixels at 16 images, 1.2 k-4.6 k at one.  On
// the 3x3 kernels' 16x16-pixel tiles they ran as 6-96 workgroups of eight sequential 64-channel chunks: 20-28 us each at
// EVERY batch size -- a tenth of the one-image step.  Here a workgroup is 64 padded rows x 64 channels (4 waves of 16
// rows), 16 KB of LDS per stage: 21-1 225 workgroups, several per CU, whose DMA latencies hide each other.
constexpr int N_BM = 64;
constexpr int N_TILE_BYTES = 64 * 128;
constexpr int N_STAGE_BYTES = 2 * N_TILE_BYTES;
constexpr int N_LDS_BYTES = 2 * N_STAGE_BYTES;

__global__ __launch_bounds__(256, 4) void conv1x1_n64_kernel(GemmArgs a) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t m0 = (int64_t)blockIdx.x * N_BM;
  const int Cin = a.Cin, Cout = a.Cout;
  const int nsteps = Cin >> 6;
  const int drow = lane >> 3, dslot = lane & 7;
  int woff[2], xoff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = (wave * 2 + i) * 8 + drow;
    const int g = (r >> 1) & 7;
    woff[i] = (r << 7) + ((dslot ^ g ^ (r & 6)) << 4);
    int64_t m = m0 + r;
    if (m >= a.Mp) m = a.Mp - 1;  // rows past the end: any valid row, never stored
    xoff[i] = (int)(m - m0) * Cin * 2 + ((dslot ^ g) << 4);
  }
  const char* wbase = reinterpret_cast<const char*>(a.wpk);
  const char* xbase = reinterpret_cast<const char*>(a.x) + m0 * Cin * 2;
  auto issue = [&](int step, int stage) {
    const char* ws = wbase + ((int64_t)step * Cout << 7);
    const char* xs = xbase + (step << 7);
    const int dst = stage * N_STAGE_BYTES + wave * 2048;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst + i * 1024), "v"(woff[i]), "s"(ws)
                   : "memory");
#pragma unroll
    for (int i = 0; i < 2; ++i)
      asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst + N_TILE_BYTES + i * 1024),
                   "v"(xoff[i]), "s"(xs)
                   : "memory");
  };
  const int t = lane & 15, q = lane >> 4, g = t >> 1;
  const int fo0 = t * 128 + (((q ^ (g & 3)) + ((g >> 2) << 2)) << 4);
  const int fo1 = t * 128 + (((q ^ (g & 3)) + (((g >> 2) ^ 1) << 2)) << 4);
  const int xfrag = N_TILE_BYTES + wave * 2048;
  f32x4 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  issue(0, 0);
  for (int step = 0; step < nsteps; ++step) {
    const int stage = step & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // this step's tiles have landed for every wave; the other stage is free again
    if (step + 1 < nsteps) issue(step + 1, stage ^ 1);
    const char* sb = smem + stage * N_STAGE_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int fo = kk ? fo1 : fo0;
      const bf16x8 xf = *reinterpret_cast<const bf16x8*>(sb + xfrag + fo);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bf16x8 wf = *reinterpret_cast<const bf16x8*>(sb + j * 2048 + fo);
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf, acc[j], 0, 0, 0);
      }
    }
  }
  const int Wp = a.W + 2, Hp = a.H + 2;
  const int cb = q * 4;
  const int m = (int)m0 + wave * 16 + t;  // Mp < 2^31 (checked by the launcher)
  const int row = m / Wp;
  const int xx = m - row * Wp;
  const int yy = row % Hp;
  if (m >= a.Mp || xx < 1 || xx > a.W || yy < 1 || yy > a.H) return;
  const int64_t off = (int64_t)m * Cout + cb;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    f32x4 v = acc[j] + *reinterpret_cast<const f32x4*>(a.bias + cb + j * 16);
    if (a.relu) {
      v.x = fmaxf(v.x, 0.f);
      v.y = fmaxf(v.y, 0.f);
      v.z = fmaxf(v.z, 0.f);
      v.w = fmaxf(v.w, 0.f);
    }
    if (a.addend != nullptr) {
      const u32x2 ad = *reinterpret_cast<const u32x2*>(a.addend + off + j * 16);
      v.x += bf16_bits_to_f32(ad.x & 0xffffu);
      v.y += __builtin_bit_cast(float, ad.x & 0xffff0000u);
      v.z += bf16_bits_to_f32(ad.y & 0xffffu);
      v.w += __builtin_bit_cast(float, ad.y & 0xffff0000u);
    }
    if (a.mask != nullptr) {
      const u32x2 mk = *reinterpret_cast<const u32x2*>(a.mask + off + j * 16);
      v.x = bf16_bits_to_f32(mk.x & 0xffffu) > 0.f ? v.x : 0.f;
      v.y = __builtin_bit_cast(float, mk.x & 0xffff0000u) > 0.f ? v.y : 0.f;
      v.z = bf16_bits_to_f32(mk.y & 0xffffu) > 0.f ? v.z : 0.f;
      v.w = __builtin_bit_cast(float, mk.y & 0xffff0000u) > 0.f ? v.w : 0.f;
    }
    *reinterpret_cast<u32x2*>(a.y + off + j * 16) = u32x2{pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w)};
  }
}

}  // namespace

// Entry for conv_mfma.hip's dispatcher (tile configuration 23).  Shapes: cin % 64 == 0, cout == 64.
int xv_launch_conv1x1_n64(const __bf16* x, const __bf16* wpk, const float* bias, __bf16* y, const __bf16* mask,
                          const __bf16* addend, int N, int H, int W, int Cin, int Cout, int relu, hipStream_t stream) {
  if ((Cin & 63) || Cout != 64) return XV_ESHAPE;
  GemmArgs a{};
  a.x = x;
  a.wpk = wpk;
  a.bias = bias;
  a.y = y;
  a.mask = mask;
  a.addend = addend;
  a.Mp = (int64_t)N * (H + 2) * (W + 2);
  a.H = H;
  a.W = W;
  a.Cin = Cin;
  a.Cout = Cout;
  a.relu = relu;
  const int64_t nblk = (a.Mp + N_BM - 1) / N_BM;
  if (nblk > 0x7fffffff || a.Mp + N_BM > 0x7fffffff || (int64_t)N_BM * Cin * 2 > 0x7fffffff) return XV_ESHAPE;
  hipLaunchKernelGGL(conv1x1_n64_kernel, dim3((unsigned)nblk), dim3(256), N_LDS_BYTES, stream, a);
  return xv_launch_status();
}

// Entry for conv_mfma.hip's dispatcher (tile configuration 18).  Shapes: cin % 64 == 0, cout % 128 == 0.
int xv_launch_conv1x1_gemm(const __bf16* x, const __bf16* wpk, const float* bias, __bf16* y, const __bf16* mask,
                           const __bf16* addend, int N, int H, int W, int Cin, int Cout, int relu, hipStream_t stream) {
  if ((Cin & 63) || (Cout & 127)) return XV_ESHAPE;
  GemmArgs a{};
  a.x = x;
  a.wpk = wpk;
  a.bias = bias;
  a.y = y;
  a.mask = mask;
  a.addend = addend;
  a.Mp = (int64_t)N * (H + 2) * (W + 2);
  a.H = H;
  a.W = W;
  a.Cin = Cin;
  a.Cout = Cout;
  a.relu = relu;
  a.n_tiles = Cout / G_BN;
  const int64_t m_tiles = (a.Mp + G_BM - 1) / G_BM;
  const int64_t nblk = m_tiles * a.n_tiles;
  if (nblk > 0x7fffffff || a.Mp + G_BM > 0x7fffffff || (int64_t)G_BM * Cin * 2 > 0x7fffffff) return XV_ESHAPE;
  a.nblk = (int)nblk;
  static bool attr_set[XV_MAX_DEVICES] = {false};
  (void)xv_allow_dynamic_lds(reinterpret_cast<const void*>(conv1x1_gemm_kernel), G_LDS_BYTES, attr_set);
  hipLaunchKernelGGL(conv1x1_gemm_kernel, dim3((unsigned)nblk), dim3(256), G_LDS_BYTES, stream, a);
  return xv_launch_status();
}
