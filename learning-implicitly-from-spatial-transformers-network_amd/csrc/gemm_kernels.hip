// MFMA kernel of the implicit MLP (network/modules.py:276-281): out = act(A . W^T + bias) with
// A [M][K] (points x features) and W [N][K] (Conv1d(k=1) weight) both bf16 hi/lo planes.
//
//   precision BF16X3: acc += A_hi.W_hi + A_hi.W_lo + A_lo.W_hi   (3 x v_mfma_f32_32x32x16_bf16,
//                     fp32 accumulate; the dropped lo.lo term is ~2^-16 relative)
//   precision BF16  : acc += A_hi.W_hi
//
// Tiling (CDNA4, wave64): workgroup 256x256 outputs, BK = 32, 8 waves as 2(M) x 4(N), each wave
// 128x64 = 4x2 tiles of 32x32 (128 accumulator registers).  Operand tiles go HBM/L2 -> LDS with
// global_load_lds_dwordx4 (no VGPR round trip), double-buffered: 4 planes x 16 KB x 2 stages =
// 128 KB LDS, one workgroup per CU.  LDS rows are 64 B (32 bf16); the 16-B chunk index is XORed
// with (row>>2)&3 -- on the SOURCE address for the linear LDS-DMA write and on the ds_read_b128
// address -- so the 16 lanes of every ds_read_b128 lane group hit 16 distinct bank quads.
// Epilogues: bias+ReLU+bf16 split store (hidden layers), fp32 store (tests), and
// bias+ReLU+dot(w3)+b3 -> sdf (fc_2 and fc_out fused; needs N == 256).
#include "list_common.h"

namespace list {

constexpr int BM = 256, BN = 256, BK = 32;
constexpr int kPlaneBytes = BM * BK * 2;        // 16 KB: one operand plane of one stage
constexpr int kStageBytes = 4 * kPlaneBytes;    // A_hi, A_lo, W_hi, W_lo
constexpr int kLdsBytes = 2 * kStageBytes;      // 128 KB

__device__ __forceinline__ void glds16(const char* g, char* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// Each wave stages two 16-row blocks (1 KB each) of every plane.  Lane i of a piece writes LDS
// bytes [16 i, 16 i + 16) of the block = (row i>>2, physical chunk i&3); it must fetch the
// LOGICAL chunk (i&3) ^ ((row>>2)&3) of that row.
template <int TERMS>
__device__ __forceinline__ void stage_tiles(const GemmParams& p, char* sbase, int m0, int n0,
                                            int kbyte, int wave, int lane) {
  const int chunk = (lane & 3) ^ ((lane >> 4) & 3);
  const int64_t ld = (int64_t)p.K * 2;
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int rb = 2 * wave + r;
    const int row = rb * 16 + (lane >> 2);
    const int64_t aoff = (int64_t)(m0 + row) * ld + kbyte + chunk * 16;
    const int64_t woff = (int64_t)(n0 + row) * ld + kbyte + chunk * 16;
    char* l = sbase + rb * 1024;
    glds16(p.a_hi + aoff, l);
    if (TERMS == 3) glds16(p.a_lo + aoff, l + kPlaneBytes);
    glds16(p.w_hi + woff, l + 2 * kPlaneBytes);
    if (TERMS == 3) glds16(p.w_lo + woff, l + 3 * kPlaneBytes);
  }
}

template <int TERMS, int EPI>
__global__ __launch_bounds__(512, 2) void k_gemm_nt(GemmParams p) {
  __shared__ __attribute__((aligned(16))) char smem[kLdsBytes];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 2, wn = wave & 3;

  // XCD-aware tile order: blocks b and b+8 share an XCD (and its L2); give each XCD a contiguous
  // run of tiles so the N-tiles of one M-tile (same A rows) run on one L2.  Bijective for any grid.
  const int tiles_n = p.N / BN;
  const int ntiles = (p.M / BM) * tiles_n;
  int tile;
  {
    const int L = blockIdx.x, q = ntiles / 8, r = ntiles % 8, xcd = L % 8;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + L / 8;
  }
  const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // fragment addresses: lane l holds A[row l&31][k = 8*(l>>5) .. +7] of each 32x16 operand block
  const int frow = lane & 31, fh = lane >> 5, swz = (lane >> 2) & 3;
  const int a_row_off = (wm * 128 + frow) * 64;
  const int w_row_off = (wn * 64 + frow) * 64;

  const int nk = p.K / BK;
  stage_tiles<TERMS>(p, smem, m0, n0, 0, wave, lane);
  __syncthreads();
  for (int t = 0; t < nk; ++t) {
    const char* cur = smem + (t & 1) * kStageBytes;
    if (t + 1 < nk)
      stage_tiles<TERMS>(p, smem + ((t + 1) & 1) * kStageBytes, m0, n0, (t + 1) * BK * 2, wave, lane);
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const int coff = ((2 * s2 + fh) ^ swz) << 4;
      bf16x8 ah[4], al[4], wh[2], wl[2];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ah[i] = *(const bf16x8*)(cur + a_row_off + i * 32 * 64 + coff);
        if (TERMS == 3) al[i] = *(const bf16x8*)(cur + kPlaneBytes + a_row_off + i * 32 * 64 + coff);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        wh[j] = *(const bf16x8*)(cur + 2 * kPlaneBytes + w_row_off + j * 32 * 64 + coff);
        if (TERMS == 3) wl[j] = *(const bf16x8*)(cur + 3 * kPlaneBytes + w_row_off + j * 32 * 64 + coff);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (TERMS == 3) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], wh[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], wl[j], acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], wh[j], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
  }

  // C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  const int col_in = lane & 31, row_in = 4 * (lane >> 5);
  if (EPI == EPI_RELU_SPLIT || EPI == EPI_F32) {
    // one base pointer per output plane; per-element offsets are (wave-uniform row term) + lane term
    const int ld = EPI == EPI_F32 ? p.N : p.ldo;
    const int64_t lane_off = (int64_t)(m0 + wm * 128 + row_in) * ld + n0 + wn * 64 + col_in;
    float* of = EPI == EPI_F32 ? p.out_f32 + lane_off : nullptr;
    unsigned short* oh = EPI == EPI_F32 ? nullptr : p.out_hi + lane_off;
    unsigned short* ol = (EPI == EPI_F32 || !p.out_lo) ? nullptr : p.out_lo + lane_off;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float bias = p.bias ? p.bias[n0 + wn * 64 + j * 32 + col_in] : 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int off = (i * 32 + (e & 3) + 8 * (e >> 2)) * ld + j * 32;
          float v = acc[i][j][e] + bias;
          if (EPI == EPI_F32) {
            if (p.relu) v = fmaxf(v, 0.f);
            of[off] = v;
          } else {
            v = fmaxf(v, 0.f);
            const unsigned short h = f2bf(v);
            oh[off] = h;
            if (ol) ol[off] = f2bf(v - bf2f(h));
          }
        }
    }
  } else {
    // sdf[row] = b3 + sum_col relu(acc + bias[col]) * w3[col]   (fc_2 + ReLU + fc_out)
    float part[4][16];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) part[i][e] = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = wn * 64 + j * 32 + col_in;      // n0 == 0 (N == BN)
      const float bias = p.bias[col], w3 = p.w3[col];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) part[i][e] += fmaxf(acc[i][j][e] + bias, 0.f) * w3;
    }
    // reduce over the 32 lanes that hold the 32 columns of a row (xor < 32 stays inside a half)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float v = part[i][e];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 8);
        v += __shfl_xor(v, 4);
        v += __shfl_xor(v, 2);
        v += __shfl_xor(v, 1);
        part[i][e] = v;
      }
    float* red = (float*)smem;                         // [4 (wn)][256 rows]; smem is idle now
    if (col_in == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int rl = wm * 128 + i * 32 + (e & 3) + 8 * (e >> 2) + row_in;
          red[wn * 256 + rl] = part[i][e];
        }
    }
    __syncthreads();
    if (threadIdx.x < 256) {
      const int row = m0 + threadIdx.x;
      if (row < p.n_valid) {
        const int rl = threadIdx.x;
        p.sdf[row] = ((red[rl] + red[256 + rl]) + (red[512 + rl] + red[768 + rl])) + p.b3[0];
      }
    }
  }
}

template <int TERMS, int EPI>
static hipError_t launch_one(const GemmParams& p, hipStream_t s) {
  const int ntiles = (p.M / BM) * (p.N / BN);
  hipLaunchKernelGGL((k_gemm_nt<TERMS, EPI>), dim3(ntiles), dim3(512), 0, s, p);
  return hipGetLastError();
}

hipError_t launch_gemm(const GemmParams& p, int terms, int epi, hipStream_t s) {
  if (p.M % BM || p.N % BN || p.K % BK || p.M <= 0) return hipErrorInvalidValue;
  if (epi == EPI_RELU_DOT && p.N != BN) return hipErrorInvalidValue;
  if (terms == 3) {
    if (epi == EPI_RELU_SPLIT) return launch_one<3, EPI_RELU_SPLIT>(p, s);
    if (epi == EPI_F32) return launch_one<3, EPI_F32>(p, s);
    return launch_one<3, EPI_RELU_DOT>(p, s);
  }
  if (epi == EPI_RELU_SPLIT) return launch_one<1, EPI_RELU_SPLIT>(p, s);
  if (epi == EPI_F32) return launch_one<1, EPI_F32>(p, s);
  return launch_one<1, EPI_RELU_DOT>(p, s);
}

}  // namespace list
