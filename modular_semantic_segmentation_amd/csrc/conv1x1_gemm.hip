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
  int pair_d1, pair_d2;  // conv1x1_gemm_wide_kernel<1>: the dilation rates of the two atrous 3x3 convs (see there)
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

// ---- wide form (round 6): 256 pixel rows x 128 channels per workgroup, three stages ---------------------------------------
// The 128x128 form above moves (128 + 128) rows x 128 B per 128x128x64 block of the product -- 64 FLOP per byte of L2
// traffic, i.e. 39 TB/s at the matrix peak: as much as the eight L2s deliver at best -- and requests a K step only one step
// (512-1 024 matrix-pipe cycles) before it is consumed: a miss to the Infinity Cache or HBM does not come back in that time.
// AdapNet's 1x1 convs (half of that expert's step, profiles/r6_adapnet_kernel_stats.csv) run at 0.3-0.8 PFLOP/s on it.
// Here: 8 waves (4 x 2, each 64 pixels x 64 channels as above), 48 KB per stage (85 FLOP per byte), THREE stages = 144 KB,
// one workgroup per CU, a K step requested two steps ahead behind a counted vmcnt (the six DMA instructions of the step in
// between stay in flight across the barrier).  Same fragment layout, same accumulation order: the same bits as the
// 128x128 form.  Where it pays is measured, not obvious (profiles/r6_conv1x1_wide_ab.txt): +30 % on the longest sums
// (24x48x9216->512: 511 -> 665 TFLOP/s), level or a few per cent either way on most shapes, -14 % on x1024->2048 alone --
// eight waves behind ONE per-step barrier hide less than two independent 4-wave workgroups per CU -- and +5 % on the whole
// AdapNet step with the launcher's rule (launches of at least three quarters of a round of workgroups).
constexpr int B_BM = 256;
constexpr int B_W_BYTES = 128 * 128, B_X_BYTES = 256 * 128;
constexpr int B_STAGE_BYTES = B_W_BYTES + B_X_BYTES;
constexpr int B_STAGES = 3;
constexpr int B_LDS_BYTES = B_STAGES * B_STAGE_BYTES;

// PAIR = true (block_b of AdapNet, adapnet.py:84-88: two atrous 3x3 convs of the same input, rates d1 / d2, concatenated): the
// same kernel as an IMPLICIT GEMM over nine taps.  The materialised form (xv_im2col_dilated_pair + this kernel over K = 18 C)
// wrote and re-read an operand 18 x the input and multiplied every output channel by the nine taps of the OTHER conv's rate
// too, whose weights are zero.  Here a workgroup's 128 output channels lie in one half of the concat (launcher: cout / 2 a
// multiple of 128), so its K loop runs over that half's nine taps only -- steps [0, 9 C/64) or [9 C/64, 18 C/64) of the same
// packed [1,1,18C,F] image -- and the pixel tile of a step is gathered by the DMA itself: row m of the tile reads padded
// pixel m + ((ky-1) (W+2) + (kx-1)) d, or, where that tap falls outside the image (or m is a border row), padded pixel 0,
// which is a border pixel of the map and therefore zero.  Same products in the same order as the materialised form (whose
// other nine taps add exact zeros): the same bits.
// MODE 2: the data gradient of the same pair (training): dx[p][cin] = sum over both convs h and their taps t of
// dy_h[p - off_h(t)] . k_h[t][cin][:] -- an implicit GEMM over 18 taps x (cout/2) channels per tap, the taps gathered from the
// two channel halves of dy with the displacement NEGATED, the weights a packed [1,1,18 cout/2,cin] image whose rows are
// (h, t, cout-of-the-half) (adapnet_trainer builds it).  a.Cin = channels of dy (both halves), a.Cout = channels of dx.
template <int MODE>
__global__ __launch_bounds__(512, 2) void conv1x1_gemm_wide_kernel(GemmArgs a) {
  constexpr bool PAIR = MODE != 0;
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  int bid = blockIdx.x;
  if ((a.nblk & 7) == 0) bid = (bid & 7) * (a.nblk >> 3) + (bid >> 3);
  const int mt = bid / a.n_tiles, nt = bid - mt * a.n_tiles;
  const int64_t m0 = (int64_t)mt * B_BM;
  const int n0 = nt * G_BN;
  const int Cin = a.Cin, Cout = a.Cout;
  const int cpt = MODE == 2 ? Cin >> 7 : Cin >> 6;            // K steps per tap (PAIR; MODE 2: of one channel half), all steps otherwise
  const int nsteps = MODE == 2 ? 18 * cpt : (PAIR ? 9 * cpt : cpt);
  const bool second = MODE == 1 && 2 * n0 >= Cout;            // this workgroup's channels belong to the second conv
  const int kbase = second ? 9 * cpt : 0;
  int dil = second ? a.pair_d2 : a.pair_d1;                   // (MODE 2: of the tap being requested, see tap_offsets)

  // DMA addressing: piece p of a tile = rows 8p .. 8p+7, lane -> (row, 16-byte slot); wave w moves weight pieces w, w + 8
  // and pixel pieces w, w + 8, w + 16, w + 24
  const int drow = lane >> 3, dslot = lane & 7;
  int woff[2], xoff[4];
  int pyy[4], pxx[4], pm[4], pslot[4];                        // PAIR: the rows' padded coordinates (yy < 0: never a valid tap)
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = (wave + 8 * i) * 8 + drow;
    const int g = (r >> 1) & 7;
    woff[i] = ((n0 + r) << 7) + ((dslot ^ g ^ (r & 6)) << 4);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (wave + 8 * i) * 8 + drow;
    const int g = (r >> 1) & 7;
    int64_t m = m0 + r;
    if (PAIR) {
      const int Wp = a.W + 2, Hp = a.H + 2;
      const int mi = (int)m, row = mi / Wp, xx = mi - row * Wp, yy = row % Hp;
      const bool interior = m < a.Mp && xx >= 1 && xx <= a.W && yy >= 1 && yy <= a.H;
      pyy[i] = interior ? yy : -(1 << 20), pxx[i] = xx, pm[i] = mi, pslot[i] = (dslot ^ g) << 4;
      xoff[i] = 0;
    } else {
      if (m >= a.Mp) m = a.Mp - 1;  // rows past the end: any valid row, never stored
      xoff[i] = (int)(m - m0) * Cin * 2 + ((dslot ^ g) << 4);
    }
  }
  const char* wbase = reinterpret_cast<const char*>(a.wpk);
  const char* xbase = reinterpret_cast<const char*>(a.x) + (PAIR ? 0 : m0 * Cin * 2);
  // PAIR: issue() is called for steps 0, 1, 2, ... in order; (itap, icb) = (step / cpt, step % cpt) follow along
  int itap = 0, icb = 0;
  auto tap_offsets = [&]() {
    const int tt = MODE == 2 && itap >= 9 ? itap - 9 : itap;
    if (MODE == 2) dil = itap >= 9 ? a.pair_d2 : a.pair_d1;
    const int ky = tt / 3, kx = tt - 3 * ky;
    const int sgn = MODE == 2 ? -1 : 1;                       // the data gradient reads dy at p - off
    const int dy = sgn * (ky - 1) * dil, dx = sgn * (kx - 1) * dil;
    const int dm = dy * (a.W + 2) + dx;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int sy = pyy[i] + dy, sx = pxx[i] + dx;
      const bool ok = sy >= 1 && sy <= a.H && sx >= 1 && sx <= a.W;
      xoff[i] = (ok ? (pm[i] + dm) * Cin * 2 : 0) + pslot[i];
    }
  };
  if (PAIR) tap_offsets();
  auto issue = [&](int step, int stage) {
    const char* ws = wbase + ((int64_t)(kbase + step) * Cout << 7);
    // MODE 2: taps 9 .. 17 read the second channel half of dy
    const char* xs = xbase + ((PAIR ? icb : step) << 7) + (MODE == 2 && itap >= 9 ? Cin : 0);
    const int dst = stage * B_STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst + (wave + 8 * i) * 1024), "v"(woff[i]), "s"(ws)
                   : "memory");
#pragma unroll
    for (int i = 0; i < 4; ++i)
      asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst + B_W_BYTES + (wave + 8 * i) * 1024),
                   "v"(xoff[i]), "s"(xs)
                   : "memory");
    if (PAIR && ++icb == cpt) {
      icb = 0;
      if (++itap < (MODE == 2 ? 18 : 9)) tap_offsets();
    }
  };

  const int t = lane & 15, q = lane >> 4, g = t >> 1;
  const int fo0 = t * 128 + (((q ^ (g & 3)) + ((g >> 2) << 2)) << 4);
  const int fo1 = t * 128 + (((q ^ (g & 3)) + (((g >> 2) ^ 1) << 2)) << 4);
  const int wfrag = wc * 64 * 128, xfrag = B_W_BYTES + wr * 64 * 128;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  issue(0, 0);
  if (nsteps > 1) issue(1, 1);
  int stage = 0;
  for (int step = 0; step < nsteps; ++step) {
    // this step's six DMA instructions have landed (the six of the next step may still be in flight) ...
    if (step + 1 < nsteps)
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // ... for every wave; and every wave is done with the stage of step - 1, which step + 2 overwrites
    const int nstage = stage == 0 ? 2 : stage - 1;  // (step + 2) % 3
    if (step + 2 < nsteps) issue(step + 2, nstage);
    const char* sb = smem + stage * B_STAGE_BYTES;
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
    stage = stage == 2 ? 0 : stage + 1;
  }

  // ---- epilogue: as the 128x128 form
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

// PAIR = true: block_b with 64 output channels (block_layer_7, 32 + 32): both convs' channels share the one 64-channel tile,
// so the K loop runs over all 18 taps of the packed [1,1,18C,64] image (the zero blocks included: the same products as the
// materialised form) -- taps 0..8 at rate d1, 9..17 at rate d2, gathered by the DMA as in conv1x1_gemm_wide_kernel<1>.
template <bool PAIR>
__global__ __launch_bounds__(256, 4) void conv1x1_n64_kernel(GemmArgs a) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t m0 = (int64_t)blockIdx.x * N_BM;
  const int Cin = a.Cin, Cout = a.Cout;
  const int cpt = Cin >> 6;                    // K steps per tap (PAIR), all steps otherwise
  const int nsteps = PAIR ? 18 * cpt : cpt;
  const int drow = lane >> 3, dslot = lane & 7;
  int woff[2], xoff[2];
  int pyy[2], pxx[2], pm[2], pslot[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = (wave * 2 + i) * 8 + drow;
    const int g = (r >> 1) & 7;
    woff[i] = (r << 7) + ((dslot ^ g ^ (r & 6)) << 4);
    int64_t m = m0 + r;
    if (PAIR) {
      const int Wp = a.W + 2, Hp = a.H + 2;
      const int mi = (int)m, row = mi / Wp, xx = mi - row * Wp, yy = row % Hp;
      const bool interior = m < a.Mp && xx >= 1 && xx <= a.W && yy >= 1 && yy <= a.H;
      pyy[i] = interior ? yy : -(1 << 20), pxx[i] = xx, pm[i] = mi, pslot[i] = (dslot ^ g) << 4;
      xoff[i] = 0;
    } else {
      if (m >= a.Mp) m = a.Mp - 1;  // rows past the end: any valid row, never stored
      xoff[i] = (int)(m - m0) * Cin * 2 + ((dslot ^ g) << 4);
    }
  }
  const char* wbase = reinterpret_cast<const char*>(a.wpk);
  const char* xbase = reinterpret_cast<const char*>(a.x) + (PAIR ? 0 : m0 * Cin * 2);
  int itap = 0, icb = 0;                       // PAIR: (step / cpt, step % cpt) of the next issue() -- steps come in order
  auto tap_offsets = [&]() {
    const int tt = itap < 9 ? itap : itap - 9, dil = itap < 9 ? a.pair_d1 : a.pair_d2;
    const int ky = tt / 3, kx = tt - 3 * ky;
    const int dy = (ky - 1) * dil, dx = (kx - 1) * dil;
    const int dm = dy * (a.W + 2) + dx;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int sy = pyy[i] + dy, sx = pxx[i] + dx;
      const bool ok = sy >= 1 && sy <= a.H && sx >= 1 && sx <= a.W;
      xoff[i] = (ok ? (pm[i] + dm) * Cin * 2 : 0) + pslot[i];   // outside the image: padded pixel 0, a zero border pixel
    }
  };
  if (PAIR) tap_offsets();
  auto issue = [&](int step, int stage) {
    const char* ws = wbase + ((int64_t)step * Cout << 7);
    const char* xs = xbase + ((PAIR ? icb : step) << 7);
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
    if (PAIR && ++icb == cpt) {
      icb = 0;
      if (++itap < 18) tap_offsets();
    }
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

// block_b's two atrous 3x3 convs + concat as ONE implicit GEMM (conv1x1_gemm_wide_kernel<1>), no materialised operand.
// wpk: the packed [1,1,18 cin,cout] image xv_im2col_dilated_pair's 1x1 conv takes (adapnet.dilated_pair_as_1x1: rows [0,9 cin)
// x columns [0,cout/2) = conv 1, rows [9 cin,18 cin) x columns [cout/2,cout) = conv 2; the other two blocks are never read).
extern "C" int xv_conv_dilated_pair_fwd(const xv_act* x, const void* wpk, const float* bias, int dilation1, int dilation2,
                                        int relu, const xv_act* y, void* stream) {
  XV_REQUIRE_BF16(x, y);
  XV_CHECK_ARG(x && y && x->data && y->data && wpk && bias);
  XV_CHECK_SHAPE(x->n == y->n && x->h == y->h && x->w == y->w && x->h > 0 && (x->c & 63) == 0 &&
                 ((y->c & 255) == 0 || y->c == 64) && dilation1 >= 1 && dilation2 >= 1);
  GemmArgs a{};
  a.x = (const __bf16*)x->data;
  a.wpk = (const __bf16*)wpk;
  a.bias = bias;
  a.y = (__bf16*)y->data;
  a.Mp = (int64_t)x->n * (x->h + 2) * (x->w + 2);
  a.H = x->h;
  a.W = x->w;
  a.Cin = x->c;
  a.Cout = y->c;
  a.relu = relu;
  a.n_tiles = y->c / G_BN;
  a.pair_d1 = dilation1;
  a.pair_d2 = dilation2;
  // 32-bit byte offsets into x from its base, tap displacement included
  const int64_t reach = (int64_t)(dilation1 > dilation2 ? dilation1 : dilation2) * (x->w + 3);
  XV_CHECK_SHAPE((a.Mp + B_BM + reach) * x->c * 2 <= 0x7fffffff);
  if (y->c == 64) {
    const int64_t nb = (a.Mp + N_BM - 1) / N_BM;
    XV_CHECK_SHAPE(nb <= 0x7fffffff);
    hipLaunchKernelGGL(conv1x1_n64_kernel<true>, dim3((unsigned)nb), dim3(256), N_LDS_BYTES, (hipStream_t)stream, a);
    return xv_launch_status();
  }
  const int64_t nblk = (a.Mp + B_BM - 1) / B_BM * a.n_tiles;
  XV_CHECK_SHAPE(nblk <= 0x7fffffff);
  a.nblk = (int)nblk;
  static bool attr_p[XV_MAX_DEVICES] = {false};
  const hipError_t e = xv_allow_dynamic_lds(reinterpret_cast<const void*>(conv1x1_gemm_wide_kernel<1>), B_LDS_BYTES, attr_p);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(conv1x1_gemm_wide_kernel<1>, dim3((unsigned)nblk), dim3(512), B_LDS_BYTES, (hipStream_t)stream, a);
  return xv_launch_status();
}

// The data gradient of xv_conv_dilated_pair_fwd (conv1x1_gemm_wide_kernel<2>).  dy [N,H,W,F] (both halves), dx [N,H,W,C];
// wpk_dgrad: packed forward-format image of the [1,1,18 F/2,C] kernel whose row (h * 9 + t) * F/2 + co holds k_h[t][:, co].
extern "C" int xv_conv_dilated_pair_bwd_data(const xv_act* dy, const void* wpk_dgrad, const float* zero_bias, int dilation1,
                                             int dilation2, const xv_act* dx, void* stream) {
  XV_REQUIRE_BF16(dy, dx);
  XV_CHECK_ARG(dy && dx && dy->data && dx->data && wpk_dgrad && zero_bias);
  XV_CHECK_SHAPE(dy->n == dx->n && dy->h == dx->h && dy->w == dx->w && dy->h > 0 && (dy->c & 127) == 0 && (dx->c & 127) == 0 &&
                 dilation1 >= 1 && dilation2 >= 1);
  GemmArgs a{};
  a.x = (const __bf16*)dy->data;
  a.wpk = (const __bf16*)wpk_dgrad;
  a.bias = zero_bias;
  a.y = (__bf16*)dx->data;
  a.Mp = (int64_t)dy->n * (dy->h + 2) * (dy->w + 2);
  a.H = dy->h;
  a.W = dy->w;
  a.Cin = dy->c;
  a.Cout = dx->c;
  a.relu = 0;
  a.n_tiles = dx->c / G_BN;
  a.pair_d1 = dilation1;
  a.pair_d2 = dilation2;
  const int64_t reach = (int64_t)(dilation1 > dilation2 ? dilation1 : dilation2) * (dy->w + 3);
  XV_CHECK_SHAPE((a.Mp + B_BM + reach) * dy->c * 2 <= 0x7fffffff);
  const int64_t nblk = (a.Mp + B_BM - 1) / B_BM * a.n_tiles;
  XV_CHECK_SHAPE(nblk <= 0x7fffffff);
  a.nblk = (int)nblk;
  static bool attr_d[XV_MAX_DEVICES] = {false};
  const hipError_t e = xv_allow_dynamic_lds(reinterpret_cast<const void*>(conv1x1_gemm_wide_kernel<2>), B_LDS_BYTES, attr_d);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(conv1x1_gemm_wide_kernel<2>, dim3((unsigned)nblk), dim3(512), B_LDS_BYTES, (hipStream_t)stream, a);
  return xv_launch_status();
}

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
  hipLaunchKernelGGL(conv1x1_n64_kernel<false>, dim3((unsigned)nblk), dim3(256), N_LDS_BYTES, stream, a);
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
  {
    // the wide form wherever the launch makes at least a quarter of a round of its workgroups (the whole AdapNet step, 16 images,
    // one box, alternating: 1 500 images/s without it, 1 574 with a three-quarter-round rule, 1 600 everywhere -- although
    // ALONE on the chip it is the slower form on most short sums, profiles/r6_conv1x1_wide_ab.txt: in the network two experts'
    // launches share the chip); XV_GEMM_WIDE=0 / 1: never / wherever the shape allows (A/B timing).  Same bits either way.
    static const int wide_env = getenv("XV_GEMM_WIDE") != nullptr ? atoi(getenv("XV_GEMM_WIDE")) : -1;
    const int64_t m_wide = (a.Mp + B_BM - 1) / B_BM;
    const int64_t nblk_wide = m_wide * a.n_tiles;
    const bool fits = nblk_wide <= 0x7fffffff && a.Mp + B_BM <= 0x7fffffff && (int64_t)B_BM * Cin * 2 <= 0x7fffffff;
    const bool enough = 4 * nblk_wide >= (int64_t)xv_num_cus();   // at least a quarter of a round: below that, more and smaller workgroups
    if (fits && wide_env != 0 && (wide_env == 1 || enough)) {
      a.nblk = (int)nblk_wide;
      static bool attr_w[XV_MAX_DEVICES] = {false};
      const hipError_t e = xv_allow_dynamic_lds(reinterpret_cast<const void*>(conv1x1_gemm_wide_kernel<0>), B_LDS_BYTES, attr_w);
      if (e != hipSuccess) return (int)e;
      hipLaunchKernelGGL(conv1x1_gemm_wide_kernel<0>, dim3((unsigned)nblk_wide), dim3(512), B_LDS_BYTES, stream, a);
      return xv_launch_status();
    }
  }
  const int64_t m_tiles = (a.Mp + G_BM - 1) / G_BM;
  const int64_t nblk = m_tiles * a.n_tiles;
  if (nblk > 0x7fffffff || a.Mp + G_BM > 0x7fffffff || (int64_t)G_BM * Cin * 2 > 0x7fffffff) return XV_ESHAPE;
  a.nblk = (int)nblk;
  static bool attr_set[XV_MAX_DEVICES] = {false};
  (void)xv_allow_dynamic_lds(reinterpret_cast<const void*>(conv1x1_gemm_kernel), G_LDS_BYTES, attr_set);
  hipLaunchKernelGGL(conv1x1_gemm_kernel, dim3((unsigned)nblk), dim3(256), G_LDS_BYTES, stream, a);
  return xv_launch_status();
}
