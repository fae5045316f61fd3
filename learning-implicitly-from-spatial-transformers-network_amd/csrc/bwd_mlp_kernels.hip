// Backward of the implicit MLP (network/modules.py:276-281 differentiated; the reference gets these
// gradients from autograd, train.py:82-85):
//
//   dZ3 = dsdf (x) w3 . [H3 > 0]                      k_head            (elementwise)
//   dW_l = dZ_l^T . A_l        (A = X, H1, H2)        k_gemm_tn         (MFMA, split over points)
//   db_l = column sums of dZ_l                        k_colsum
//   dZ_{l-1} = (dZ_l . W_l) . [H_{l-1} > 0]           k_gemm_nt + EPI_MASK_SPLIT (gemm_kernels.hip)
//   dX = dZ1 . W0                                     k_gemm_nt + EPI_DX
//
// k_gemm_tn contracts over the POINT index, which is the row index of both operands (dZ [P][M] and
// X [P][N] are stored point-major).  The tiles are therefore staged as they lie in memory, [p][column],
// by LDS-DMA, and the MFMA operands -- 8 consecutive p for one column per lane -- are read with
// ds_read_b64_tr_b16, gfx950's transposing LDS read (a 4 x 16 block of 16-bit elements delivered
// column-major to a 16-lane group), so no transposed copy of the 1.2-GB feature matrix is ever made.
// LDS image: 512-B rows (256 columns), 16-B chunk index XORed with ((row&3)<<2 | (row>>2)&3): the four
// rows of a transposed read and the chunks of its two blocks fall on 16 distinct 16-B bank slots.
#include <stdlib.h>
#include <string.h>

#include "list_common.h"

namespace list {

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

constexpr int kTnLds = 131072;

template <int TERMS> struct TnPipe {
  static constexpr int kPlanes = TERMS == 3 ? 4 : 2;
  static constexpr int BK = TERMS == 3 ? 32 : 64;               // points per stage
  static constexpr int kRowBytes = 512;                         // 256 columns x 16 bit
  static constexpr int kPlaneBytes = BK * kRowBytes;            // 16 or 32 KB
  static constexpr int kStageBytes = kPlanes * kPlaneBytes;     // 64 KB, two stages
  static constexpr int kPiecesPerWave = BK / 2 / 8;             // 1-KB pieces (2 rows) per plane per wave
  static constexpr int kBOff = (TERMS == 3 ? 2 : 1) * kPlaneBytes;
};

__device__ __forceinline__ int tn_swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

__device__ __forceinline__ void glds16_tn(const char* g, char* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

__device__ __forceinline__ s16x4 tr_read(const char* l) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)l);
}

// lane i of a piece writes LDS bytes [16 i, 16 i + 16) of the piece = (row i / 32, physical chunk
// i % 32) and fetches the logical chunk phys ^ swz(row) (low 4 bits) of that row.
template <int TERMS>
__device__ __forceinline__ void stage_tn(const GemmTnParams& p, char* sbase, int64_t prow0, int m0,
                                         int n0, int wave, int lane) {
  using P = TnPipe<TERMS>;
#pragma unroll
  for (int r = 0; r < P::kPiecesPerWave; ++r) {
    const int piece = P::kPiecesPerWave * wave + r;
    const int row = piece * 2 + (lane >> 5);
    const int phys = lane & 31;
    const int logical = (phys & 16) | ((phys & 15) ^ tn_swz(row));
    const int acol = m0 + logical * 8;
    int bcol = n0 + logical * 8;
    if (bcol > p.N - 8) bcol = p.N - 8;          // partial last tile: re-read valid columns (outputs unused)
    const int64_t aoff = ((prow0 + row) * (int64_t)p.lda + acol) * 2;
    // b_x3i: B is the feature matrix X with its hi / lo halfs interleaved in 64-B blocks (list_common.h xi_off)
    const int64_t boff = p.b_x3i ? xi_off((prow0 + row) * (int64_t)p.ldb + bcol) * 2
                                 : ((prow0 + row) * (int64_t)p.ldb + bcol) * 2;
    char* l = sbase + piece * 1024;
    glds16_tn(p.a_hi + aoff, l);
    if (TERMS == 3) glds16_tn(p.a_lo + aoff, l + P::kPlaneBytes);
    glds16_tn(p.b_hi + boff, l + P::kBOff);
    if (TERMS == 3) glds16_tn((p.b_x3i ? p.b_hi + 2 * kXiLo : p.b_lo) + boff, l + P::kBOff + P::kPlaneBytes);
  }
}

template <int FP16>
__device__ __forceinline__ f32x16 mfma_tn(const s16x8& a, const s16x8& b, const f32x16& c) {
  if (FP16)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b),
                                                  c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b),
                                                 c, 0, 0, 0);
}

__device__ __forceinline__ s16x8 cat4(const s16x4& lo, const s16x4& hi) {
  return (s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// grid = splits * tiles; workgroup 256 x 256 outputs, 8 waves as 2 (M) x 4 (N), wave 128 x 64.
//
// Register cap (round 3).  Left to itself the compiler spends the whole budget of two waves per SIMD (250 of 256
// registers, most of them on hoisted fragment reads), and two such waves own the SIMD's 512 registers: no wave of
// another kernel is ever resident beside a weight-gradient workgroup.  In the forked backward dW0 runs beside the
// voxel scatters and the map-side gathers for most of their time (list_capi.hip), kernels of 16 - 40 registers per
// wave that wait on memory: capped at 194 registers (no scratch; the kernel itself is not slower, 0.78 ms) two of its
// waves leave room for four of theirs per SIMD, and the fp16 backward takes 4.50 instead of 4.73 ms.
// amdgpu_num_vgpr(112) is the LIMIT handed to the register allocator (architectural registers; the backend doubles it
// for the unified file: at most 224); what the compiler then USES is 194, in all three instantiations the attribute
// covers -- <1, 1> fp16, <1, 0> plain bf16, <3, 0> bf16x3 under LIST_TN_SHAPE=32 -- each with ScratchSize 0
// (-Rpass-analysis=kernel-resource-usage, round 4).  The next lower limits that compile (96 and below) spill.
template <int TERMS, int FP16>
__global__ __launch_bounds__(512, 2) __attribute__((amdgpu_num_vgpr(112))) void k_gemm_tn(GemmTnParams p) {
  using P = TnPipe<TERMS>;
  __shared__ __attribute__((aligned(16))) char smem[kTnLds];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int tiles_n = (p.N + 255) / 256;
  const int ntiles = (p.M / 256) * tiles_n;
  const int tile = blockIdx.x % ntiles, split = blockIdx.x / ntiles;
  const int m0 = (tile / tiles_n) * 256, n0 = (tile % tiles_n) * 256;
  const int nk = p.P / P::BK;
  const int t0 = split * p.steps_per_split;
  const int t1 = min(nk, t0 + p.steps_per_split);

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // transposed-read addresses: group g = lane>>4 covers columns 16 (g&1) .. +15 and points 8 (g>>1) .. +7
  // of a 32-column x 16-point operand block; lane 4q+pp of the group addresses row q, columns 4pp .. 4pp+3
  const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
  int a_off[2][4], b_off[2][2];
#pragma unroll
  for (int hh = 0; hh < 2; ++hh) {
    const int rowb = 8 * (g >> 1) + 4 * hh + q;
    const int f = (q << 2) | ((2 * (g >> 1) + hh) & 3);                  // tn_swz(16 s + rowb)
    const int low = 2 * (g & 1) + (pp >> 1);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      a_off[hh][i] = 512 * rowb + 16 * (16 * wm + ((4 * i + low) ^ f)) + 8 * (pp & 1);
#pragma unroll
    for (int j = 0; j < 2; ++j)
      b_off[hh][j] = P::kBOff + 512 * rowb + 16 * (16 * (wn >> 1) + ((8 * (wn & 1) + 4 * j + low) ^ f)) + 8 * (pp & 1);
  }

  if (t0 < t1) stage_tn<TERMS>(p, smem, (int64_t)t0 * P::BK, m0, n0, wave, lane);
  for (int t = t0; t < t1; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (t + 1 < t1)
      stage_tn<TERMS>(p, smem + ((t + 1 - t0) & 1) * P::kStageBytes, (int64_t)(t + 1) * P::BK, m0, n0, wave, lane);
    const char* cur = smem + ((t - t0) & 1) * P::kStageBytes;
    // register double-buffered fragments: the transposed reads of k16-step s2+1 are issued ahead of the
    // MFMAs of step s2 (same scheme as k_gemm_nt)
    constexpr int NS = P::BK / 16;
    s16x8 ah[2][4], al[2][4], bh[2][2], bl[2][2];
    auto load_frags = [&](int s2, int buf) {
      const char* cs = cur + s2 * 16 * P::kRowBytes;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ah[buf][i] = cat4(tr_read(cs + a_off[0][i]), tr_read(cs + a_off[1][i]));
        if (TERMS == 3)
          al[buf][i] = cat4(tr_read(cs + P::kPlaneBytes + a_off[0][i]), tr_read(cs + P::kPlaneBytes + a_off[1][i]));
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        bh[buf][j] = cat4(tr_read(cs + b_off[0][j]), tr_read(cs + b_off[1][j]));
        if (TERMS == 3)
          bl[buf][j] = cat4(tr_read(cs + P::kPlaneBytes + b_off[0][j]), tr_read(cs + P::kPlaneBytes + b_off[1][j]));
      }
    };
    load_frags(0, 0);
#pragma unroll
    for (int s2 = 0; s2 < NS; ++s2) {
      const int bb = s2 & 1;
      if (s2 + 1 < NS) load_frags(s2 + 1, bb ^ 1);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (TERMS == 3) {
            acc[i][j] = mfma_tn<0>(al[bb][i], bh[bb][j], acc[i][j]);
            acc[i][j] = mfma_tn<0>(ah[bb][i], bl[bb][j], acc[i][j]);
          }
          acc[i][j] = mfma_tn<FP16>(ah[bb][i], bh[bb][j], acc[i][j]);
        }
    }
  }

  // C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  const int col_in = lane & 31, row_in = 4 * (lane >> 5);
  float* out = p.slab + ((int64_t)split * p.M + m0 + wm * 128 + row_in) * p.ldn + n0 + wn * 64 + col_in;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    if (n0 + wn * 64 + j * 32 + col_in >= p.N) continue;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e)
        out[(int64_t)(i * 32 + (e & 3) + 8 * (e >> 2)) * p.ldn + j * 32] = acc[i][j][e];
  }
}

// ---- the same product on the 16 x 16 x 32 MFMA shape (round 3) ------------------------------------------------------
// Same tiles, same LDS image, same staging and the same number of transposed reads as k_gemm_tn; the chip holds a
// higher clock under twice as many half-size MFMAs (the round-2b finding on fc_0).  Operand
// layout of v_mfma_f32_16x16x32: lane l holds 8 consecutive k (= points) 8 (l >> 4) .. + 7 of row / column l & 15, so
// the 16-lane group g = l >> 4 reads points 8 g + 4 hh + q (two transposed reads hh = 0, 1 of 4 points each) of ONE
// 16-column block: lane 4 q + pp of the group addresses point row q, columns 4 pp .. 4 pp + 3.  With the image's chunk
// swizzle tn_swz(row) = (q << 2) | ((2 g + hh) & 3) the 16 (g, q) pairs of one read fall on every pair of 16-B bank
// slots exactly twice (f >> 1 = 2 q + (g & 1)): two passes per 512-B read, the minimum.
template <int FP16>
__device__ __forceinline__ f32x4 mfma_tn16(const s16x8& a, const s16x8& b, const f32x4& c) {
  if (FP16)
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <int TERMS, int FP16>
__global__ __launch_bounds__(512, 2) void k_gemm_tn16(GemmTnParams p) {
  using P = TnPipe<TERMS>;
  __shared__ __attribute__((aligned(16))) char smem[kTnLds];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int tiles_n = (p.N + 255) / 256;
  const int ntiles = (p.M / 256) * tiles_n;
  const int tile = blockIdx.x % ntiles, split = blockIdx.x / ntiles;
  const int m0 = (tile / tiles_n) * 256, n0 = (tile % tiles_n) * 256;
  const int nk = p.P / P::BK;
  const int t0 = split * p.steps_per_split;
  const int t1 = min(nk, t0 + p.steps_per_split);

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;

  const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
  // byte offset of (point row 8 g + 4 hh + q, swizzle f) inside a 32-point block; the 16-B chunk of column block c
  // (0 .. 15 inside the wave's half of the row) is ((2 c + (pp >> 1)) ^ f) -- one XOR per read, no offset tables
  int rowoff[2], fz[2];
#pragma unroll
  for (int hh = 0; hh < 2; ++hh) {
    rowoff[hh] = 512 * (8 * g + 4 * hh + q) + 8 * (pp & 1);
    fz[hh] = (q << 2) | ((2 * g + hh) & 3);
  }
  const int alow = pp >> 1, abase = 256 * wm;                       // A: columns 128 wm + 16 i: chunks 16 wm + 2 i
  const int bbase = P::kBOff + 256 * (wn >> 1), bch = 8 * (wn & 1);  // B: columns 64 wn + 16 j: chunks 8 wn + 2 j

  if (t0 < t1) stage_tn<TERMS>(p, smem, (int64_t)t0 * P::BK, m0, n0, wave, lane);
  for (int t = t0; t < t1; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (t + 1 < t1)
      stage_tn<TERMS>(p, smem + ((t + 1 - t0) & 1) * P::kStageBytes, (int64_t)(t + 1) * P::BK, m0, n0, wave, lane);
    const char* cur = smem + ((t - t0) & 1) * P::kStageBytes;
    constexpr int NS = P::BK / 32;
    // register double-buffered fragments, as in k_gemm_tn: the transposed reads of k32-step s2 + 1 are issued ahead of
    // the MFMAs of step s2 (single-plane formats: two steps per K-tile; the split formats have one)
    s16x8 ah[2][8], bh[2][4], al[TERMS == 3 ? 8 : 1], bl[TERMS == 3 ? 4 : 1];
    auto load_frags = [&](int s2, int buf) {
      const char* cs = cur + s2 * 32 * P::kRowBytes;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int o0 = bbase + rowoff[0] + 16 * ((bch + 2 * j + alow) ^ fz[0]);
        const int o1 = bbase + rowoff[1] + 16 * ((bch + 2 * j + alow) ^ fz[1]);
        bh[buf][j] = cat4(tr_read(cs + o0), tr_read(cs + o1));
        if (TERMS == 3) bl[j] = cat4(tr_read(cs + P::kPlaneBytes + o0), tr_read(cs + P::kPlaneBytes + o1));
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int o0 = abase + rowoff[0] + 16 * ((2 * i + alow) ^ fz[0]);
        const int o1 = abase + rowoff[1] + 16 * ((2 * i + alow) ^ fz[1]);
        ah[buf][i] = cat4(tr_read(cs + o0), tr_read(cs + o1));
        if (TERMS == 3) al[i] = cat4(tr_read(cs + P::kPlaneBytes + o0), tr_read(cs + P::kPlaneBytes + o1));
      }
    };
    load_frags(0, 0);
#pragma unroll
    for (int s2 = 0; s2 < NS; ++s2) {
      const int bb = s2 & 1;
      if (s2 + 1 < NS) load_frags(s2 + 1, bb ^ 1);
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (TERMS == 3) {
            acc[i][j] = mfma_tn16<0>(al[i], bh[bb][j], acc[i][j]);
            acc[i][j] = mfma_tn16<0>(ah[bb][i], bl[j], acc[i][j]);
          }
          acc[i][j] = mfma_tn16<FP16>(ah[bb][i], bh[bb][j], acc[i][j]);
        }
    }
  }

  // C/D layout of the 16x16 MFMA: col = lane & 15, row = 4 (lane >> 4) + reg
  const int col_in = lane & 15, row_in = 4 * (lane >> 4);
  float* out = p.slab + ((int64_t)split * p.M + m0 + wm * 128 + row_in) * p.ldn + n0 + wn * 64 + col_in;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (n0 + wn * 64 + j * 16 + col_in >= p.N) continue;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) out[(int64_t)(i * 16 + e) * p.ldn + j * 16] = acc[i][j][e];
  }
}

int wgrad_splits(int M, int N, int P, int terms) {
  const int nk = P / (terms == 3 ? 32 : 64);
  const int s = wgrad_nominal_splits(M, N);
  return s > nk ? (nk < 1 ? 1 : nk) : s;
}

hipError_t launch_gemm_tn(const GemmTnParams& p, int terms, hipStream_t s) {
  if (p.M <= 0 || p.M % 256 || p.N < 8 || p.N % 8 || p.P <= 0 || p.P % 256 || p.splits < 1)
    return hipErrorInvalidValue;
  const int ntiles = (p.M / 256) * ((p.N + 255) / 256);
  const dim3 grid((unsigned)(ntiles * p.splits));
  // MFMA shape per operand format, from interleaved A/B runs of the training step: the
  // single-plane formats keep 32 x 32 x 16 (k_gemm_tn: 194 registers under its cap; the 16 x 16 x 32 form needs 246, is
  // 5 % slower on its own and leaves no room beside it); the split formats -- three MFMAs per product, one k32-step per
  // K-tile -- take 16 x 16 x 32 (212 registers; 0.1 ms ahead of the capped 32 x 32 x 16 form in the bf16x3 step).
  // LIST_TN_SHAPE=16 / 32 forces one (A/B runs).
  static const int forced = [] { const char* e = getenv("LIST_TN_SHAPE"); return e ? atoi(e) : 0; }();
  const bool shape32 = forced == 32 || (forced != 16 && terms != 3);
  if (shape32) {
    if (p.fmt == FMT_FP16) hipLaunchKernelGGL((k_gemm_tn<1, 1>), grid, dim3(512), 0, s, p);
    else if (terms == 3) hipLaunchKernelGGL((k_gemm_tn<3, 0>), grid, dim3(512), 0, s, p);
    else hipLaunchKernelGGL((k_gemm_tn<1, 0>), grid, dim3(512), 0, s, p);
  } else {
    if (p.fmt == FMT_FP16) hipLaunchKernelGGL((k_gemm_tn16<1, 1>), grid, dim3(512), 0, s, p);
    else if (terms == 3) hipLaunchKernelGGL((k_gemm_tn16<3, 0>), grid, dim3(512), 0, s, p);
    else hipLaunchKernelGGL((k_gemm_tn16<1, 0>), grid, dim3(512), 0, s, p);
  }
  return hipGetLastError();
}

// ---- slab reduction -------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_wgrad_reduce(const float* __restrict__ slab, int splits, int M, int N,
                                                      int ldn, FeatLayout L, int use_layout,
                                                      const float* __restrict__ scale, float* __restrict__ out,
                                                      int ldo) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)M * N) return;
  const int m = (int)(i / N), n = (int)(i - (int64_t)m * N);
  int col = n;
  if (use_layout) {
    col = ref_index_of(L, n);
    if (col < 0) return;
  }
  const float* src = slab + (int64_t)m * ldn + n;
  const int64_t stride = (int64_t)M * ldn;
  float acc = 0.f;
  for (int s = 0; s < splits; ++s) acc += src[s * stride];     // fixed order: bitwise reproducible
  out[(int64_t)m * ldo + col] = scale ? acc * scale[1] : acc;
}

hipError_t launch_wgrad_reduce(const float* slab, int splits, int M, int N, int ldn, const FeatLayout* L,
                               const float* scale, float* out, int ldo, hipStream_t s) {
  FeatLayout dummy;
  memset(&dummy, 0, sizeof(dummy));
  const int64_t total = (int64_t)M * N;
  hipLaunchKernelGGL(k_wgrad_reduce, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, slab, splits, M, N,
                     ldn, L ? *L : dummy, L ? 1 : 0, scale, out, ldo);
  return hipGetLastError();
}

// ---- gradient scale (FP16 operands) and d(fc_out.bias) -------------------------------------------------------
// s = 2^k with max|dsdf| * s in [32, 64): every gradient operand of the chain is linear in dsdf, so the
// fp16 range is centred on the data and the fp32 epilogues multiply by 1/s.  s = 1 for bf16 operands.
// Two launches (per-workgroup partials, then one workgroup): fixed summation order, reproducible.
__global__ __launch_bounds__(256) void k_grad_scale_partial(const float* __restrict__ g, int64_t n,
                                                            float* __restrict__ partial) {
  __shared__ float smax[256], ssum[256];
  float mx = 0.f, sum = 0.f;
  const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (i0 + e < n) { const float v = g[i0 + e]; mx = fmaxf(mx, fabsf(v)); sum += v; }
  smax[threadIdx.x] = mx; ssum[threadIdx.x] = sum;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) {
      smax[threadIdx.x] = fmaxf(smax[threadIdx.x], smax[threadIdx.x + off]);
      ssum[threadIdx.x] += ssum[threadIdx.x + off];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { partial[2 * blockIdx.x] = smax[0]; partial[2 * blockIdx.x + 1] = ssum[0]; }
}

__global__ __launch_bounds__(256) void k_grad_scale_final(const float* __restrict__ partial, int nb, int fp16,
                                                          float* __restrict__ scale, float* __restrict__ db3) {
  __shared__ float smax[256], ssum[256];
  float mx = 0.f, sum = 0.f;
  for (int i = threadIdx.x; i < nb; i += 256) { mx = fmaxf(mx, partial[2 * i]); sum += partial[2 * i + 1]; }
  smax[threadIdx.x] = mx; ssum[threadIdx.x] = sum;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) {
      smax[threadIdx.x] = fmaxf(smax[threadIdx.x], smax[threadIdx.x + off]);
      ssum[threadIdx.x] += ssum[threadIdx.x + off];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    float s = 1.f;
    const float m = smax[0];
    if (fp16 && m > 0.f && m < 3.0e38f) {
      int e;
      (void)frexpf(m, &e);                      // m = f * 2^e, f in [0.5, 1)
      int k = 6 - e;                            // m * 2^k in [32, 64)
      k = k > 100 ? 100 : (k < -100 ? -100 : k);
      s = ldexpf(1.f, k);
    }
    scale[0] = s; scale[1] = 1.f / s; scale[2] = ssum[0]; scale[3] = 0.f;
    if (db3) db3[0] = ssum[0];
  }
}

hipError_t launch_grad_scale(const float* grad_sdf, int64_t n, int fp16, float* scale, float* db3,
                             float* partial, hipStream_t s) {
  const int nb = (int)((n + 1023) / 1024);
  hipLaunchKernelGGL(k_grad_scale_partial, dim3((unsigned)nb), dim3(256), 0, s, grad_sdf, n, partial);
  hipLaunchKernelGGL(k_grad_scale_final, dim3(1), dim3(256), 0, s, partial, nb, fp16, scale, db3);
  return hipGetLastError();
}

// ---- dZ3[r][n] = s * dsdf[point of row r] * w3[n] * [H3[r][n] > 0]; padded rows are zero ---------------
template <int FMT>
__global__ __launch_bounds__(256) void k_head(const float* __restrict__ grad_sdf, const int* __restrict__ order,
                                              int n_valid, int rows, int H3, const unsigned short* __restrict__ h3,
                                              const float* __restrict__ w3, const float* __restrict__ scale,
                                              unsigned short* __restrict__ dz_hi, unsigned short* __restrict__ dz_lo) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= (int64_t)rows * H3) return;
  const int r = (int)(i / H3), n = (int)(i - (int64_t)r * H3);
  float gs = 0.f;
  if (r < n_valid) gs = grad_sdf[order ? order[r] : r] * scale[0];
  const uint2 hm = *(const uint2*)(h3 + i);
  const float4 w = *(const float4*)(w3 + n);
  float4 v;
  v.x = (hm.x & 0x7fffu) ? gs * w.x : 0.f;
  v.y = (hm.x & 0x7fff0000u) ? gs * w.y : 0.f;
  v.z = (hm.y & 0x7fffu) ? gs * w.z : 0.f;
  v.w = (hm.y & 0x7fff0000u) ? gs * w.w : 0.f;
  if (FMT == FMT_FP16) {
    *(uint2*)(dz_hi + i) = half4(v);
  } else {
    uint2 hi, lo;
    split4(v, hi, lo);
    *(uint2*)(dz_hi + i) = hi;
    if (dz_lo) *(uint2*)(dz_lo + i) = lo;
  }
}

hipError_t launch_head(const float* grad_sdf, const int* order, int n_valid, int rows, int H3,
                       const unsigned short* h3_hi, const float* w3, const float* scale,
                       unsigned short* dz_hi, unsigned short* dz_lo, int fmt, hipStream_t s) {
  const int64_t total = (int64_t)rows * H3 / 4;
  const dim3 grid((unsigned)((total + 255) / 256));
  if (fmt == FMT_FP16)
    hipLaunchKernelGGL(k_head<FMT_FP16>, grid, dim3(256), 0, s, grad_sdf, order, n_valid, rows, H3, h3_hi, w3,
                       scale, dz_hi, dz_lo);
  else
    hipLaunchKernelGGL(k_head<FMT_BF16_SPLIT>, grid, dim3(256), 0, s, grad_sdf, order, n_valid, rows, H3, h3_hi,
                       w3, scale, dz_hi, dz_lo);
  return hipGetLastError();
}

// ---- column sums (bias gradients, d fc_out.weight) ------------------------------------------------------------
// One workgroup per kColsumRows rows: a thread owns 8 consecutive columns (one 16-B load per plane and
// row), N/8 threads cover a row, the rest of the workgroup takes further rows in parallel; LDS combines
// the row lanes -> partial[chunk][N].  A second launch adds the chunks in order (bitwise reproducible).
__device__ __forceinline__ void decode8(const uint4& h, int fmt_fp16, float (&f)[8]) {
  const unsigned w[4] = {h.x, h.y, h.z, h.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    f[2 * e] = fmt_fp16 ? h2f((unsigned short)(w[e] & 0xffff)) : bf2f((unsigned short)(w[e] & 0xffff));
    f[2 * e + 1] = fmt_fp16 ? h2f((unsigned short)(w[e] >> 16)) : bf2f((unsigned short)(w[e] >> 16));
  }
}

template <int FMT>
__global__ __launch_bounds__(256) void k_colsum(const unsigned short* __restrict__ z_hi,
                                                const unsigned short* __restrict__ z_lo, int rows, int n_valid,
                                                int N, const float* __restrict__ grad_sdf,
                                                const int* __restrict__ order, float* __restrict__ partial) {
  __shared__ float red[2048];
  const int tpr = N / 8;                               // threads per row (<= 256)
  const int lanes = 256 / tpr;                         // rows in flight
  const int rl = threadIdx.x / tpr, cq = threadIdx.x - rl * tpr;
  const int r0 = blockIdx.x * kColsumRows;
  const int r1 = min(rows, r0 + kColsumRows);
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  if (rl < lanes) {
    for (int r = r0 + rl; r < r1; r += lanes) {
      const int64_t o = (int64_t)r * N + cq * 8;
      float f[8];
      decode8(*(const uint4*)(z_hi + o), FMT == FMT_FP16, f);
      if (FMT != FMT_FP16 && z_lo) {
        float l[8];
        decode8(*(const uint4*)(z_lo + o), 0, l);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] += l[e];
      }
      float wr = 1.f;
      if (grad_sdf) wr = r < n_valid ? grad_sdf[order ? order[r] : r] : 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = fmaf(wr, f[e], acc[e]);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[rl * N + cq * 8 + e] = acc[e];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < N; c += 256) {
    float sum = 0.f;
    for (int l = 0; l < lanes; ++l) sum += red[l * N + c];
    partial[(int64_t)blockIdx.x * N + c] = sum;
  }
}

// grid = N / 32; block = 32 columns x 8 chunk lanes
__global__ __launch_bounds__(256) void k_colsum_final(const float* __restrict__ partial, int chunks, int N,
                                                      const float* __restrict__ scale, int use_inv_scale,
                                                      float* __restrict__ out) {
  __shared__ float red[8][32];
  const int c = blockIdx.x * 32 + (threadIdx.x & 31), kl = threadIdx.x >> 5;
  float acc = 0.f;
  if (c < N)
    for (int k = kl; k < chunks; k += 8) acc += partial[(int64_t)k * N + c];
  red[kl][threadIdx.x & 31] = acc;
  __syncthreads();
  if (kl == 0 && c < N) {
    float sum = 0.f;
#pragma unroll
    for (int l = 0; l < 8; ++l) sum += red[l][threadIdx.x];
    out[c] = use_inv_scale ? sum * scale[1] : sum;
  }
}

hipError_t launch_colsum(const unsigned short* z_hi, const unsigned short* z_lo, int rows, int n_valid,
                         int N, int fmt, const float* grad_sdf, const int* order, const float* scale,
                         int use_inv_scale, float* partial, float* out, hipStream_t s) {
  if (N % 8 || N / 8 > 256 || N > 2048) return hipErrorInvalidValue;
  const int chunks = (rows + kColsumRows - 1) / kColsumRows;
  if (fmt == FMT_FP16)
    hipLaunchKernelGGL(k_colsum<FMT_FP16>, dim3((unsigned)chunks), dim3(256), 0, s, z_hi, z_lo, rows, n_valid, N,
                       grad_sdf, order, partial);
  else
    hipLaunchKernelGGL(k_colsum<FMT_BF16_SPLIT>, dim3((unsigned)chunks), dim3(256), 0, s, z_hi, z_lo, rows, n_valid,
                       N, grad_sdf, order, partial);
  hipLaunchKernelGGL(k_colsum_final, dim3((unsigned)((N + 31) / 32)), dim3(256), 0, s, partial, chunks, N,
                     scale, use_inv_scale, out);
  return hipGetLastError();
}

// ---- transposed weight copies ------------------------------------------------------------------------------
// wt[k][n] = W[n][src(k)] as 16-bit planes; src = ref_index_of for fc_0 (gather-order rows, zero padding)
__global__ __launch_bounds__(256) void k_prep_wt(const float* __restrict__ w, int Nout, int Kin_ref, int Krows,
                                                 FeatLayout L, int use_layout, int fmt,
                                                 unsigned short* __restrict__ hi, unsigned short* __restrict__ lo) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)Krows * Nout) return;
  const int k = (int)(i / Nout), n = (int)(i - (int64_t)k * Nout);
  int src = k;
  if (use_layout) src = (k < L.Kp) ? ref_index_of(L, k) : -1;
  const float v = (src >= 0 && src < Kin_ref) ? w[(int64_t)n * Kin_ref + src] : 0.f;
  if (fmt == FMT_FP16) {
    hi[i] = f2h(v);
  } else {
    const unsigned short h = f2bf(v);
    hi[i] = h;
    lo[i] = bf_lo(v, h);
  }
}

hipError_t launch_prep_weights_bwd(const ListMlpWeights& w, const FeatLayout& L, const PackedMlpBwd& P,
                                   char* packed, hipStream_t s) {
  const int fmt = w.precision == LIST_PREC_FP16 ? FMT_FP16 : FMT_BF16_SPLIT;
  FeatLayout dummy;
  memset(&dummy, 0, sizeof(dummy));
  int order = 0;          // three transposed copies, three parts of `packed`: only the first keeps the stream's order
  auto go = [&](const float* src, int Nout, int Kin_ref, int Krows, bool use_layout, size_t hi, size_t lo) {
    const int64_t total = (int64_t)Krows * Nout;
    LIST_LAUNCH(k_prep_wt, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, order, src, Nout, Kin_ref,
                Krows, use_layout ? L : dummy, use_layout ? 1 : 0, fmt, (unsigned short*)(packed + hi),
                (unsigned short*)(packed + lo));
    order = any_order();
  };
  go(w.w0, w.H1, w.F, P.KpT, true, P.w0t_hi, P.w0t_lo);
  go(w.w1, w.H2, w.H1, w.H1, false, P.w1t_hi, P.w1t_lo);
  go(w.w2, w.H3, w.H2, w.H2, false, P.w2t_hi, P.w2t_lo);
  return hipGetLastError();
}

}  // namespace list
