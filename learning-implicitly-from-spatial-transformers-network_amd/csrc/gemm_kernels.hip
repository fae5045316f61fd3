// MFMA kernels of the implicit MLP (network/modules.py:276-281): out = act(A . W^T + bias) with
// A [M][K] (points x features) and W [N][K] (Conv1d(k=1) weight) as bf16 hi/lo planes or one fp16 plane.
//
//   precision BF16X3: acc += A_lo.W_hi + A_hi.W_lo + A_hi.W_hi   (3 MFMAs per operand pair, fp32
//                     accumulate; the dropped lo.lo term is ~2^-16 relative)
//   precision BF16  : acc += A_hi.W_hi          precision FP16: acc += A.W (fp16 operands)
//
// Kernels: k_gemm_nt (plain 2-stage loop, 32x32x16 MFMA), k_gemm_nt16 (the same on 16x16x32), and
// k_gemm_nt_pp (ping-pong schedule; from 8 K-tiles on; single-plane operands, or the split formats with their hi /
// lo halfs interleaved in 64-B blocks -- fc_0's X and packed weight, the projection of the perceptual map).
// Tiling (CDNA4, wave64): workgroup 256x256 outputs, 8 waves as 2(M) x 4(N), each wave 128x64 =
// 4x2 tiles of 32x32 or 8x4 tiles of 16x16 (128 accumulator registers).  Operand tiles go HBM/L2 -> LDS with
// global_load_lds_dwordx4 (no VGPR round trip), double-buffered 64-KB stages (128 KB LDS, one
// workgroup per CU).  LDS rows are 64 B (BK = 32) or 128 B (BK = 64); the 16-B chunk index is XORed
// with (row>>2)&3 resp. (row>>1)&7 -- on the SOURCE address for the linear LDS-DMA write and on the
// ds_read_b128 address -- so the 16 lanes of every ds_read_b128 lane group hit 16 distinct 16-B
// slots of the 256-B bank row.
// Epilogues: bias+ReLU+bf16 split store (hidden layers), fp32 store (tests), and
// bias+ReLU+dot(w3)+b3 -> sdf (fc_2 and fc_out fused; needs N == 256).
#include <type_traits>

#include "list_common.h"
#include "mfma_common.h"

namespace list {

constexpr int BM = 256, BN = 256;
constexpr int kLdsBytes = 131072;               // 128 KB of the CU's 160 KB

// Software pipeline.  A stage = the operand planes of one K-step:
//   TERMS = 3 (bf16 hi/lo): {A_hi, A_lo, W_hi, W_lo}, BK = 32 (64-B rows),  4 x 16 KB, 2 stages;
//   TERMS = 1 (fp16/bf16) : {A, W},                   BK = 64 (128-B rows), 2 x 32 KB, 2 stages.
// With one plane per operand the K-step is 64 so that every LDS-DMA row segment is a full 128-B
// line: 64-B half-line requests cap the L2 -> LDS rate near 30 GB/s per CU (measured: the load path
// alone took 0.60 ms of fc_0's 0.64 ms), full lines roughly double it.
// LDS-DMA loads run kAhead K-steps ahead; the only waits in the loop are a COUNTED s_waitcnt vmcnt
// (the wave's own pieces of the oldest stage have landed, younger stages stay in flight) followed by
// ONE raw s_barrier (every wave's pieces have landed, and every wave is done reading the buffer
// about to be refilled).
template <int TERMS> struct Pipe {
  static constexpr int kPlanes = TERMS == 3 ? 4 : 2;
  static constexpr int BK = TERMS == 3 ? 32 : 64;
  static constexpr int kRowBytes = BK * 2;                      // 64 or 128
  static constexpr int kPlaneBytes = BM * kRowBytes;            // 16 or 32 KB
  static constexpr int kStageBytes = kPlanes * kPlaneBytes;     // 64 KB
  static constexpr int kStages = kLdsBytes / kStageBytes;       // 2
  static constexpr int kAhead = kStages - 1;                    // 1
  static constexpr int kRowsPerPiece = 1024 / kRowBytes;        // rows moved by one 1-KB LDS-DMA
  static constexpr int kPiecesPerWave = BM / kRowsPerPiece / 8; // per plane per wave: 2 or 4
  static constexpr int kLoadsPerStage = kPiecesPerWave * kPlanes;
  static constexpr int kWOff = (TERMS == 3 ? 2 : 1) * kPlaneBytes;
  // XOR swizzle of the 16-B chunk index inside a row (conflict-free ds_read_b128, see header)
  __device__ static __forceinline__ int swz(int row) {
    return kRowBytes == 64 ? ((row >> 2) & 3) : ((row >> 1) & 7);
  }
};

// Each wave stages kPiecesPerWave 1-KB pieces of every plane.  Lane i of a piece writes LDS bytes
// [16 i, 16 i + 16) of the piece = (row i / chunks_per_row, physical chunk i % chunks_per_row); it
// must fetch the LOGICAL chunk phys ^ swz(row) of that row (the LDS image stays lane-linear, the
// swizzle lives in the per-lane SOURCE address and in the ds_read address).
template <int TERMS>
__device__ __forceinline__ void stage_tiles(const GemmParams& p, char* sbase, int m0, int n0,
                                            int kbyte, int wave, int lane) {
  using P = Pipe<TERMS>;
  constexpr int CPR = P::kRowBytes / 16;                 // chunks per row: 4 or 8
  const int64_t ld = (int64_t)p.K * 2;
#pragma unroll
  for (int r = 0; r < P::kPiecesPerWave; ++r) {
    const int piece = P::kPiecesPerWave * wave + r;
    const int row = piece * P::kRowsPerPiece + lane / CPR;
    const int chunk = (lane % CPR) ^ P::swz(row);
    const int64_t aoff = (int64_t)(m0 + row) * ld + kbyte + chunk * 16;
    const int64_t woff = (int64_t)(n0 + row) * ld + kbyte + chunk * 16;
    char* l = sbase + piece * 1024;
#ifndef LIST_GEMM_NO_A
    glds16(p.a_hi + aoff, l);
    if (TERMS == 3) glds16(p.a_lo + aoff, l + P::kPlaneBytes);
#endif
#ifndef LIST_GEMM_NO_W
    glds16(p.w_hi + woff, l + P::kWOff);
    if (TERMS == 3) glds16(p.w_lo + woff, l + P::kWOff + P::kPlaneBytes);
#endif
  }
}

template <int FP16>
__device__ __forceinline__ f32x16 mfma(const bf16x8& a, const bf16x8& b, const f32x16& c) {
  if (FP16)   // same registers, fp16 interpretation: v_mfma_f32_32x32x16_f16
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a),
                                                  __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
template <int FP16>
__device__ __forceinline__ unsigned short to_half_plane(float v) { return FP16 ? f2h(v) : f2bf(v); }

// ---- staged epilogue --------------------------------------------------------------------------------------
// The MFMA accumulator layout gives a lane one COLUMN of its tile (32 lanes = 32 consecutive columns of
// one row): storing 16-bit outputs from it means 2-byte stores in 64-B runs, 128 of them per lane, and the
// same again for every per-element input (ReLU mask).  Instead each wave turns its 128 x 64 tile through
// LDS, 32 rows at a time (the operand stages are dead by then), and every lane gets 8 consecutive columns
// of a row: one 16-B load per input plane, one 16-B store per output plane, 128-B runs per row.

template <typename F>
__device__ __forceinline__ void staged_epilogue(const f32x16 (&acc)[4][2], char* smem, int wave, int lane,
                                                F&& emit) {
  float* tile = (float*)smem + wave * (32 * kStageLd);
  const int col_in = lane & 31, row_in = 4 * (lane >> 5);
  const int rr = lane >> 3, c8 = (lane & 7) * 8;
  __syncthreads();                                 // every wave is done with the operand stages
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e)
        tile[((e & 3) + 8 * (e >> 2) + row_in) * kStageLd + j * 32 + col_in] = acc[i][j][e];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int r = rr + 8 * k;
      const float4 a = *(const float4*)(tile + r * kStageLd + c8);
      const float4 b = *(const float4*)(tile + r * kStageLd + c8 + 4);
      const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
      emit(i * 32 + r, c8, v);
    }
    __syncthreads();
  }
}

// ---- epilogues of the 32x32 kernels ------------------------------------------------------------------------
template <int EPI, int FP16>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, f32x16 (&acc)[4][2], char* smem, int m0, int n0,
                                              int wm, int wn, int wave, int lane) {
  // C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  const int col_in = lane & 31, row_in = 4 * (lane >> 5);
  if (EPI == EPI_MASK_SPLIT) {
    // ReLU backward: dZ_prev = acc where the saved activation is positive (network/modules.py:276-278)
    const int64_t row_base = m0 + wm * 128;
    const int col_base = n0 + wn * 64;
    staged_epilogue(acc, smem, wave, lane, [&](int r, int c8, const float (&v)[8]) {
      const int64_t row = row_base + r;
      const uint4 mk = *(const uint4*)(p.mask + row * p.ldmask + col_base + c8);
      const unsigned mw[4] = {mk.x, mk.y, mk.z, mk.w};
      float o[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o[2 * e] = (mw[e] & 0x7fffu) ? v[2 * e] : 0.f;
        o[2 * e + 1] = (mw[e] & 0x7fff0000u) ? v[2 * e + 1] : 0.f;
      }
      store8_planes<FP16>(p.out_hi, p.out_lo, row * p.ldo + col_base + c8, o);
    });
  } else if (EPI == EPI_DX) {
    const int64_t row_base = m0 + wm * 128;
    const int col_base = n0 + wn * 64;
    staged_epilogue(acc, smem, wave, lane, [&](int r, int c8, const float (&v)[8]) {
      if (col_base + c8 >= p.n_store) return;                  // n_store is a multiple of 8
      const int64_t off = (row_base + r) * p.ldo + col_base + c8;
      if (p.dx_f16) {
        const uint2 a = half4(make_float4(v[0], v[1], v[2], v[3])), b = half4(make_float4(v[4], v[5], v[6], v[7]));
        *(uint4*)((unsigned short*)p.dx + off) = make_uint4(a.x, a.y, b.x, b.y);
      } else {
        *(float4*)((float*)p.dx + off) = make_float4(v[0], v[1], v[2], v[3]);
        *(float4*)((float*)p.dx + off + 4) = make_float4(v[4], v[5], v[6], v[7]);
      }
    });
  } else if (EPI == EPI_RELU_SPLIT) {
    const int64_t row_base = m0 + wm * 128;
    const int col_base = n0 + wn * 64;
    bool bad = false;
    staged_epilogue(acc, smem, wave, lane, [&](int r, int c8, const float (&v)[8]) {
      float o[8];
      const float4 b0 = p.bias ? *(const float4*)(p.bias + col_base + c8) : make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 b1 = p.bias ? *(const float4*)(p.bias + col_base + c8 + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) { o[e] = relu_nan(v[e] + bb[e]); bad = bad || (o[e] != o[e]); }
      store8_planes<FP16>(p.out_hi, p.out_lo, (row_base + r) * p.ldo + col_base + c8, o);
    });
    // a NaN in a feature row reaches every output column of that row: flag the row tile for the exact redo of
    // its gathers (gather_kernels.hip: exact border semantics)
    if (p.nan_tiles && __builtin_amdgcn_ballot_w64(bad) != 0 && lane == 0) p.nan_tiles[m0 / BM] = 1;
  } else if (EPI == EPI_RELU_SPLIT || EPI == EPI_F32) {
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
            if (p.relu) v = relu_nan(v);
            of[off] = v;
          } else {
            v = relu_nan(v);
            const unsigned short h = to_half_plane<FP16>(v);
            oh[off] = h;
            if (!FP16 && ol) ol[off] = bf_lo(v, h);
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
        for (int e = 0; e < 16; ++e) part[i][e] += relu_nan(acc[i][j][e] + bias) * w3;
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
    __syncthreads();                                   // every wave is done reading the last stage
    float* red = (float*)smem;                         // [4 (wn)][256 rows]
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
        p.sdf[p.order ? p.order[row] : row] =
            ((red[rl] + red[256 + rl]) + (red[512 + rl] + red[768 + rl])) + p.b3[0];
      }
    }
  }
}

// TERMS = 3: bf16 hi/lo split (3 products); TERMS = 1: single plane, bf16 (FP16 = 0) or fp16 (FP16 = 1)
template <int TERMS, int EPI, int FP16>
__global__ __launch_bounds__(512, 2) void k_gemm_nt(GemmParams p) {
  __shared__ __attribute__((aligned(16))) char smem[kLdsBytes];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 2, wn = wave & 3;

  // XCD-aware tile order: blocks b and b+8 share an XCD (and its L2); give each XCD a contiguous
  // run of tiles so the N-tiles of one M-tile (same A rows) run on one L2.  Bijective for any grid.
  const int tiles_n = p.N / BN;
  const int ntiles = (p.M / BM) * tiles_n;
  const int tile = xcd_contiguous_block(blockIdx.x, ntiles);
  const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
  if (p.tile_gate && p.tile_gate[tile / tiles_n] == 0) return;      // gated re-run (exact border semantics): uniform exit

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // fragment addresses: lane l holds A[row l&31][k = 8*(l>>5) .. +7] of each 32x16 operand block
  using P = Pipe<TERMS>;
  const int frow = lane & 31, fh = lane >> 5;
  const int fswz = P::swz(frow);                       // tile row offsets are multiples of 32
  const int a_row_off = (wm * 128 + frow) * P::kRowBytes;
  const int w_row_off = (wn * 64 + frow) * P::kRowBytes;

  const int nk = p.K / P::BK;
#pragma unroll
  for (int s = 0; s < P::kAhead; ++s)
    if (s < nk) stage_tiles<TERMS>(p, smem + s * P::kStageBytes, m0, n0, s * P::kRowBytes, wave, lane);
  for (int t = 0; t < nk; ++t) {
    // stage t must have landed; stages t+1 .. t+kAhead-1 (those that exist) may stay in flight
    const int younger = min(P::kAhead - 1, nk - 1 - t);
    if (younger >= 2) wait_vmcnt<2 * P::kLoadsPerStage>();
    else if (younger == 1) wait_vmcnt<P::kLoadsPerStage>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
#ifndef LIST_GEMM_NO_LOAD
    if (t + P::kAhead < nk)
      stage_tiles<TERMS>(p, smem + ((t + P::kAhead) % P::kStages) * P::kStageBytes, m0, n0,
                         (t + P::kAhead) * P::kRowBytes, wave, lane);
#endif
    const char* cur = smem + (t % P::kStages) * P::kStageBytes;
    // Register double-buffered fragments: the ds_reads of k16-step s2+1 are issued ahead of the MFMAs
    // of step s2 (written as one buffer, hipcc reuses four fragment registers and exposes an
    // lgkmcnt(0) round trip every four MFMAs; sched_group_barrier pinning measured no further gain).
    constexpr int NS = P::BK / 16;
    bf16x8 ah[2][4], al[2][4], wh[2][2], wl[2][2];
    auto load_frags = [&](int s2, int buf) {
      const int coff = ((2 * s2 + fh) ^ fswz) << 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ah[buf][i] = *(const bf16x8*)(cur + a_row_off + i * 32 * P::kRowBytes + coff);
        if (TERMS == 3)
          al[buf][i] = *(const bf16x8*)(cur + P::kPlaneBytes + a_row_off + i * 32 * P::kRowBytes + coff);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        wh[buf][j] = *(const bf16x8*)(cur + P::kWOff + w_row_off + j * 32 * P::kRowBytes + coff);
        if (TERMS == 3)
          wl[buf][j] = *(const bf16x8*)(cur + P::kWOff + P::kPlaneBytes + w_row_off + j * 32 * P::kRowBytes + coff);
      }
    };
#ifdef LIST_GEMM_NO_FRAGS
    continue;        // ablation: LDS-DMA stream + barrier only
#endif
    load_frags(0, 0);
#pragma unroll
    for (int s2 = 0; s2 < NS; ++s2) {
      const int b = s2 & 1;
      if (s2 + 1 < NS) load_frags(s2 + 1, b ^ 1);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (TERMS == 3) {
            acc[i][j] = mfma<0>(al[b][i], wh[b][j], acc[i][j]);
            acc[i][j] = mfma<0>(ah[b][i], wl[b][j], acc[i][j]);
          }
#ifdef LIST_GEMM_NO_MFMA
          asm volatile("" ::"v"(ah[b][i]), "v"(wh[b][j]));
#else
          acc[i][j] = mfma<FP16>(ah[b][i], wh[b][j], acc[i][j]);
#endif
        }
    }
  }

  gemm_epilogue<EPI, FP16>(p, acc, smem, m0, n0, wm, wn, wave, lane);
}

// ---- the 16x16x32 MFMA shape ----------------------------------------------------------------------------------
// Identical tiling, staging and pipeline; the wave's 128x64 outputs are 8x4 tiles of 16x16 (4 accumulator
// registers each).  Operand lane map: lane l holds A[row l&15][k = 8*(l>>4) .. +7] of a 16x32 block; C/D:
// col = lane&15, row = 4*(lane>>4) + reg.  MI355X holds a higher clock on this shape under sustained MFMA load
// (fc_0, same schedule, same box: 1.98 GHz and 968 k cycles against 1.84 GHz and 1003 k cycles on 32x32x16).
// staged epilogue of the 16x16 accumulator layout: 32 rows (two 16-row tiles) x 64 columns per turn; the 64
// lanes of every ds_write_b32 hit 64 distinct banks (row stride 68 floats: +4 rows = +16 banks)
template <typename F>
__device__ __forceinline__ void staged_epilogue16(const f32x4v (&acc)[8][4], char* smem, int wave, int lane,
                                                  F&& emit) {
  float* tile = (float*)smem + wave * (32 * kStageLd);
  const int col_in = lane & 15, row_in = 4 * (lane >> 4);
  const int rr = lane >> 3, c8 = (lane & 7) * 8;
  __syncthreads();                                 // every wave is done with the operand stages
#pragma unroll
  for (int c = 0; c < 4; ++c) {
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          tile[(ii * 16 + row_in + e) * kStageLd + j * 16 + col_in] = acc[2 * c + ii][j][e];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int r = rr + 8 * k;
      const float4 a = *(const float4*)(tile + r * kStageLd + c8);
      const float4 b = *(const float4*)(tile + r * kStageLd + c8 + 4);
      const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
      emit(c * 32 + r, c8, v);
    }
    __syncthreads();
  }
}

// epilogues of the 16x16 kernels: hidden layer (bias + ReLU + 16-bit planes, NaN probe), fp32 (diagnostic),
// fused fc_2 + fc_out
// dx_group: EPI_DX output of a grouped launch's product (null: p.dx)
template <int EPI, int FP16>
__device__ __forceinline__ void gemm_epilogue16(const GemmParams& p, f32x4v (&acc)[8][4], char* smem, int m0, int n0,
                                                int wm, int wn, int wave, int lane, void* dx_group = nullptr) {
  const int col_in = lane & 15, row_in = 4 * (lane >> 4);
  if (EPI == EPI_RELU_SPLIT) {
    const int64_t row_base = m0 + wm * 128;
    const int col_base = n0 + wn * 64;
    bool bad = false;
    if (p.rowvec) {
      // the perceptual block's contribution, sampled from the projected map into the head of every X row: the same
      // turns as staged_epilogue16, with the row vectors of turn c + 1 requested while turn c is processed (they
      // come from HBM with a 7-KB row stride; fetched where they are used they cost a dependent round trip per
      // 32 rows: +14 us per tile)
      float* tile = (float*)smem + wave * (32 * kStageLd);
      const int rr = lane >> 3, c8 = (lane & 7) * 8;
      const float4 b0 = p.bias ? *(const float4*)(p.bias + col_base + c8) : make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 b1 = p.bias ? *(const float4*)(p.bias + col_base + c8 + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
      float4 rv[2][4][2];
      auto load_rv = [&](int c, int buf) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float* src = (const float*)(p.rowvec + (row_base + c * 32 + rr + 8 * k) * p.rowvec_stride) + col_base + c8;
          rv[buf][k][0] = *(const float4*)src;
          rv[buf][k][1] = *(const float4*)(src + 4);
        }
      };
      load_rv(0, 0);
      __syncthreads();                                 // every wave is done with the operand stages
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (c + 1 < 4) load_rv(c + 1, (c + 1) & 1);
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e)
              tile[(ii * 16 + row_in + e) * kStageLd + j * 16 + col_in] = acc[2 * c + ii][j][e];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int r = rr + 8 * k;
          const float4 a = *(const float4*)(tile + r * kStageLd + c8);
          const float4 b = *(const float4*)(tile + r * kStageLd + c8 + 4);
          const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
          const float4 r0 = rv[c & 1][k][0], r1 = rv[c & 1][k][1];
          const float ad[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
          float o[8];
          // (acc + row vector) + bias: the K sum first, like the reference's fc_0
#pragma unroll
          for (int e = 0; e < 8; ++e) { o[e] = relu_nan((v[e] + ad[e]) + bb[e]); bad = bad || (o[e] != o[e]); }
          store8_planes<FP16>(p.out_hi, p.out_lo, (row_base + c * 32 + r) * p.ldo + col_base + c8, o);
        }
        __syncthreads();
      }
    } else
    staged_epilogue16(acc, smem, wave, lane, [&](int r, int c8, const float (&v)[8]) {
      float o[8];
      const float4 b0 = p.bias ? *(const float4*)(p.bias + col_base + c8) : make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 b1 = p.bias ? *(const float4*)(p.bias + col_base + c8 + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) { o[e] = relu_nan(v[e] + bb[e]); bad = bad || (o[e] != o[e]); }
      store8_planes<FP16>(p.out_hi, p.out_lo, (row_base + r) * p.ldo + col_base + c8, o);
    });
    // NaN probe for the exact redo of the row tile's gathers (see gemm_epilogue)
    if (p.nan_tiles && __builtin_amdgcn_ballot_w64(bad) != 0 && lane == 0) p.nan_tiles[m0 / BM] = 1;
  } else if (EPI == EPI_MASK_SPLIT) {
    const int64_t row_base = m0 + wm * 128;
    const int col_base = n0 + wn * 64;
    staged_epilogue16(acc, smem, wave, lane, [&](int r, int c8, const float (&v)[8]) {
      const int64_t row = row_base + r;
      const uint4 mk = *(const uint4*)(p.mask + row * p.ldmask + col_base + c8);
      const unsigned mw[4] = {mk.x, mk.y, mk.z, mk.w};
      float o[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o[2 * e] = (mw[e] & 0x7fffu) ? v[2 * e] : 0.f;
        o[2 * e + 1] = (mw[e] & 0x7fff0000u) ? v[2 * e + 1] : 0.f;
      }
      store8_planes<FP16>(p.out_hi, p.out_lo, row * p.ldo + col_base + c8, o);
    });
  } else if (EPI == EPI_DX) {
    const int64_t row_base = m0 + wm * 128;
    const int col_base = n0 + wn * 64;
    void* const dx = dx_group ? dx_group : p.dx;
    staged_epilogue16(acc, smem, wave, lane, [&](int r, int c8, const float (&v)[8]) {
      if (col_base + c8 >= p.n_store) return;                  // n_store is a multiple of 8
      const int64_t off = (row_base + r) * p.ldo + col_base + c8;
      if (p.dx_f16) {
        const uint2 a = half4(make_float4(v[0], v[1], v[2], v[3])), b = half4(make_float4(v[4], v[5], v[6], v[7]));
        *(uint4*)((unsigned short*)dx + off) = make_uint4(a.x, a.y, b.x, b.y);
      } else {
        *(float4*)((float*)dx + off) = make_float4(v[0], v[1], v[2], v[3]);
        *(float4*)((float*)dx + off + 4) = make_float4(v[4], v[5], v[6], v[7]);
      }
    });
  } else if (EPI == EPI_F32) {
    const int64_t lane_off = (int64_t)(m0 + wm * 128 + row_in) * p.N + n0 + wn * 64 + col_in;
    float* of = p.out_f32 + lane_off;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float bias = p.bias ? p.bias[n0 + wn * 64 + j * 16 + col_in] : 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = acc[i][j][e] + bias;
          if (p.relu) v = relu_nan(v);
          of[(i * 16 + e) * p.N + j * 16] = v;
        }
    }
  } else {
    float part[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) part[i][e] = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = wn * 64 + j * 16 + col_in;      // n0 == 0 (N == BN)
      const float bias = p.bias[col], w3 = p.w3[col];
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) part[i][e] += relu_nan(acc[i][j][e] + bias) * w3;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = part[i][e];
        v += __shfl_xor(v, 8);
        v += __shfl_xor(v, 4);
        v += __shfl_xor(v, 2);
        v += __shfl_xor(v, 1);
        part[i][e] = v;
      }
    if (p.out_hi) {
      // a forward that keeps its activations: H3 = relu(fc_2) leaves as 16-bit planes like H1 / H2 (the backward's
      // head takes its ReLU mask, dZ3 and d fc_out.weight from it; until round 3 it re-ran this product for them).
      // Same values the dot product above used; the staged helper brackets its LDS turns with barriers.
      const int64_t row_base = m0 + wm * 128;
      const int col_base = wn * 64;                    // n0 == 0 (N == BN)
      staged_epilogue16(acc, smem, wave, lane, [&](int r, int c8, const float (&v)[8]) {
        const float4 b0 = *(const float4*)(p.bias + col_base + c8), b1 = *(const float4*)(p.bias + col_base + c8 + 4);
        const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = relu_nan(v[e] + bb[e]);
        store8_planes<FP16>(p.out_hi, p.out_lo, (row_base + r) * p.ldo + col_base + c8, o);
      });
    }
    __syncthreads();
    float* red = (float*)smem;                         // [4 (wn)][256 rows]
    if (col_in == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[wn * 256 + wm * 128 + i * 16 + row_in + e] = part[i][e];
    }
    __syncthreads();
    if (threadIdx.x < 256) {
      const int row = m0 + threadIdx.x;
      if (row < p.n_valid) {
        const int rl = threadIdx.x;
        p.sdf[p.order ? p.order[row] : row] =
            ((red[rl] + red[256 + rl]) + (red[512 + rl] + red[768 + rl])) + p.b3[0];
      }
    }
  }
}

// ---- ping-pong schedule (single-plane operands, BK = 64) -----------------------------------------------------------
// Same tile, LDS image, swizzle and epilogues as the plain loops; a different schedule, after the guide's 256^2 8-phase
// template.  A K-tile is four phases, one 64 x 32 quadrant of the wave's 128 x 64 outputs each (over K = 64: 16 MFMAs
// of 16x16x32, or 8 of 32x32x16 with -DLIST_PP_SHAPE32):
//   p0: read A(first 64 rows), W(first 32 cols) -> q(0,0)   p1: read W(second) -> q(0,1)   p2: read A(second) -> q(1,1)
//   p3: -> q(1,0)
// Every phase is [READ: ds_reads + ONE quarter of the next K-tile's LDS-DMA staging | s_barrier | MFMAs | s_barrier],
// and the two wave groups (wm = 0 / 1: the two waves of every SIMD) run ONE barrier apart, so that on each SIMD one
// wave reads LDS and issues loads while the other issues MFMAs.  The MFMA cluster is held between its two barriers
// by an s_setprio pair and scheduling fences: without them hipcc moves MFMAs across the raw s_barrier (they touch
// registers only) in among the next READ section and the groups stop alternating (round 2: fc_0 0.605 -> 0.567 ms).
// Staging runs kLead phases ahead of the phase that owns the quarter (quarter g = 4 tile + {A first halves, W first
// halves, W second halves, A second halves} is issued in phase g - kLead) and the wait is a counted
// vmcnt(2 (kLead - 2)) at the end of every READ section: the kLead - 2 youngest quarters stay in flight across the
// barriers, the loop never drains to 0.
// What bounds it (round-2 ablations on fc_0, one box): MFMAs + barriers alone 0.42 ms (the pipe idles ~95
// cycles per barrier interval), the LDS-DMA stream alone 0.42 ms (0.29 ms with A resident in L2), both 0.55 ms; never
// waiting for the loads, or kLead 5 / 6, changes nothing: the per-CU rate of the vector-memory path (~57 cycles per
// 1-KB piece with A from HBM) paces the loop, in shader cycles -- hence the 16x16x32 shape, on which the chip holds
// 1.98 instead of 1.84 GHz: 0.55 -> 0.49 ms.
// Ordering: every wave has waited for its loads of quarter g by phase g-2, the barrier behind the later group's
// wait makes it complete for all, and it is first read in phase g-1 at the earliest (W first halves; A first g,
// W second g-1, A second g-1).  The 8 quarter slots are the two K-tile buffers; a slot is re-staged two phases
// after its last read or later.
__device__ __forceinline__ int a_quarter_row(int q, int half) { return (q >> 3) * 128 + half * 64 + (q & 7) * 8; }
__device__ __forceinline__ int w_quarter_row(int q, int half) { return (q >> 2) * 64 + half * 32 + (q & 3) * 8; }

#ifndef LIST_PP_LEAD
#define LIST_PP_LEAD 4
#endif
constexpr int kLead = LIST_PP_LEAD;      // phases between the issue of a staging quarter and the phase that owns it (4..6)
#ifdef LIST_PP_SHAPE32
constexpr bool kPpShape16 = false;
#else
constexpr bool kPpShape16 = true;
#endif
template <int P> __device__ __forceinline__ void pp_prio() {
#ifndef LIST_PP_NO_SETPRIO
  __builtin_amdgcn_s_setprio(P);
#endif
}
__device__ __forceinline__ void pp_fence() {
#ifndef LIST_PP_NO_FENCE
  __builtin_amdgcn_sched_barrier(0);
#endif
}

// X3 (split formats, 16x16x32 shape only): the operands hold hi and lo halfs interleaved in 64-B blocks (xi_off), so a
// 128-B LDS row is [32 hi | 32 lo] of a K-tile of 32 columns -- the staging, the LDS image and the fragment reads are
// those of the single-plane kernel (k-step 0 = the hi halfs, k-step 1 = the lo halfs); only the MFMA section differs:
// X3 == 3: acc += a_lo.w_hi + a_hi.w_lo + a_hi.w_hi (the order of the plain split kernel: bit-identical to it);
// X3 == 1: a_hi.w_hi only (plain bf16 precision on the interleaved operands).
template <int EPI, int FP16, bool S16, int X3 = 0>
__global__ __launch_bounds__(512, 2) void k_gemm_nt_pp(GemmParams p) {
  using P = Pipe<1>;
  static_assert(X3 == 0 || (S16 && !FP16), "interleaved split operands: 16x16x32 bf16 only");
  static_assert(P::BK == 64 && P::kRowBytes == 128, "single-plane pipeline");
  __shared__ __attribute__((aligned(16))) char smem[kLdsBytes];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int tiles_n = p.N / BN;
  const int ntiles = (p.M / BM) * tiles_n;
  const int tile = xcd_contiguous_block(blockIdx.x, ntiles);
  int m0 = (tile / tiles_n) * BM;
  const int n0 = (tile % tiles_n) * BN;
  if (p.tile_gate && p.tile_gate[tile / tiles_n] == 0) return;      // gated re-run (exact border semantics): uniform exit
  // operands of this tile: the launch's, or (a grouped EPI_DX launch) those of the product the row tile belongs to --
  // scalar selects over the kernel arguments at constant offsets (no indexed copy of the argument block)
  const char* A = p.a_hi; const char* Wt = p.w_hi;
  int Kk = p.K, lda_e = p.lda ? p.lda : p.K, a_rows_e = p.a_rows ? p.a_rows : p.M;
  [[maybe_unused]] void* dx_out = p.dx;
  if constexpr (EPI == EPI_DX) {
    if (p.n_groups > 0) {
      int mt = tile / tiles_n;
      bool found = false;
#pragma unroll
      for (int i = 0; i < kGemmMaxGroups; ++i) {
        if (!found && i < p.n_groups) {
          if (mt < p.grp[i].m_tiles || i + 1 == p.n_groups) {
            A = p.grp[i].a; Wt = p.grp[i].w; dx_out = p.grp[i].out; Kk = p.grp[i].K; a_rows_e = p.grp[i].a_rows;
            lda_e = p.grp[i].lda; found = true;
          } else {
            mt -= p.grp[i].m_tiles;
          }
        }
      }
      m0 = mt * BM;
    }
  }

  // accumulators: 4x2 tiles of 32x32 (16 registers each) or 8x4 tiles of 16x16 (4 each); 128 registers either way
  typedef typename std::conditional<S16, f32x4v, f32x16>::type acc_t;
  constexpr int TI = S16 ? 8 : 4, TJ = S16 ? 4 : 2, TE = S16 ? 4 : 16;
  acc_t acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int e = 0; e < TE; ++e) acc[i][j][e] = 0.f;

  // operand blocks: 32 rows x 16 k (lane: row l&31, k-octet l>>5) or 16 rows x 32 k (row l&15, k-octet l>>4)
  constexpr int TR = S16 ? 16 : 32;            // rows of an operand block
  constexpr int KS = S16 ? 2 : 4;              // MFMA k-steps per K-tile
  constexpr int KO = S16 ? 4 : 2;              // 16-B k-octets per MFMA k-step
  constexpr int HA = 64 / TR, HW = 32 / TR;    // operand blocks per A half (64 rows) / W half (32 columns)
  const int frow = lane & (TR - 1), fo = lane / TR;
  const int fswz = P::swz(frow);               // block row offsets are multiples of 16
  const int a_row_off = (wm * 128 + frow) * P::kRowBytes;
  const int w_row_off = (wn * 64 + frow) * P::kRowBytes;
  const int64_t lda = (int64_t)lda_e * (X3 ? 4 : 2);      // bytes per operand row
  const int64_t ldw = (int64_t)(p.ldw ? p.ldw : p.K) * (X3 ? 4 : 2);
  const int a_last = a_rows_e - 1;                         // rows beyond the last existing one re-read it (outputs unused)
  const int nk = X3 ? Kk / 32 : Kk / P::BK;                // K-tiles of 128 operand bytes per row

  // one staging quarter = 16 pieces of 1 KB (8 rows of 128 B); this wave moves pieces 2 wave and 2 wave + 1
  auto stage_quarter = [&](char* sbase, int kbyte, int quarter) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int q = 2 * wave + r;
      const bool is_a = quarter == 0 || quarter == 3;
      const int row0 = is_a ? a_quarter_row(q, quarter == 3) : w_quarter_row(q, quarter == 2);
      const int row = row0 + lane / 8;
      const int chunk = (lane % 8) ^ P::swz(row);
#ifdef LIST_PP_A_RESIDENT     // ablation (wrong results): every tile stages the A rows of the first M-tile -- no HBM stream for A,
      // the LDS-DMA volume of a 128 x 512 tile that streams W twice and produces its A on chip
      const char* g = (is_a ? A + (int64_t)min(row, a_last) * lda : Wt + (int64_t)(n0 + row) * ldw) + kbyte + chunk * 16;
#else
      const char* g = (is_a ? A + (int64_t)min(m0 + row, a_last) * lda : Wt + (int64_t)(n0 + row) * ldw) + kbyte + chunk * 16;
#endif
#ifndef LIST_PP_A_DEFAULT_POLICY
      if (EPI == EPI_RELU_SPLIT && is_a) glds16_nt(g, sbase + row0 * P::kRowBytes);
      else
#endif
      glds16(g, sbase + (is_a ? 0 : P::kWOff) + row0 * P::kRowBytes);
    }
  };
  auto frag = [&](const char* cur, int plane_off, int row_off, int blk, int ks) -> bf16x8 {
    return *(const bf16x8*)(cur + plane_off + row_off + blk * TR * P::kRowBytes + (((KO * ks + fo) ^ fswz) << 4));
  };

  // byte offset of K-tile t in the operand rows: tiles from k_gap_at on lie k_gap tiles further (fc_0 without the projected
  // levels of its perceptual block, list_prep_img_proj); scalar arithmetic, 0 / 0 = no gap
  const int gap_at = p.k_gap_at, gap = p.k_gap;
  auto ktb = [&](int t) { return (t + (t >= gap_at ? gap : 0)) * P::kRowBytes; };
  // prologue: quarters 0 .. kLead-1; the first K-tile must be complete for everybody; then the second wave
  // group falls one barrier behind
#pragma unroll
  for (int g = 0; g < kLead; ++g)
    if (g / 4 < nk) stage_quarter(smem + ((g / 4) & 1) * P::kStageBytes, ktb(g / 4), g & 3);
  if (nk > 1) wait_vmcnt<2 * (kLead - 4)>(); else wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  if (wm == 1) __builtin_amdgcn_s_barrier();
  const int last_quarter = 4 * nk - 1;

  bf16x8 a[HA][KS], w[2][HW][KS];     // A blocks of the current 64-row half; W blocks of both 32-column halves
  // one K-tile; STEADY: every phase still has kLead - 2 younger quarters in flight (no branches in the body)
  auto k_tile = [&](int t, auto steady) {
    constexpr bool STEADY = decltype(steady)::value;
    const char* cur = smem + (t & 1) * P::kStageBytes;
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) {
      // ---- READ section: this phase's fragments, one staging quarter kLead phases ahead, the counted wait
      if (ph == 0 || ph == 1) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
          for (int b = 0; b < HW; ++b) w[ph][b][ks] = frag(cur, P::kWOff, w_row_off, ph * HW + b, ks);
      }
      if (ph == 0 || ph == 2) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
          for (int b = 0; b < HA; ++b) a[b][ks] = frag(cur, 0, a_row_off, (ph / 2) * HA + b, ks);
      }
      // quarter g = k + kLead of the sequence (k = 4 t + ph)
      const int tt = t + (ph + kLead) / 4;
      const int sq = (ph + kLead) & 3;
#ifdef LIST_PP_NO_LOAD        // ablation: only the prologue's quarters are ever staged (real data, no load stream)
      if (false) {
#else
      if (STEADY) {
#endif
        stage_quarter(smem + (tt & 1) * P::kStageBytes, ktb(tt), sq);
        wait_vmcnt<2 * (kLead - 2)>();     // my loads of every quarter up to k + 2 have landed
      } else {
#ifndef LIST_PP_NO_LOAD
        if (tt < nk) stage_quarter(smem + (tt & 1) * P::kStageBytes, ktb(tt), sq);
#endif
        const int beyond = last_quarter - (4 * t + ph + 2);        // quarters issued beyond k + 2
        if (beyond >= 4) wait_vmcnt<2 * (kLead - 2 < 4 ? kLead - 2 : 4)>();
        else if (beyond == 3) wait_vmcnt<2 * (kLead - 2 < 3 ? kLead - 2 : 3)>();
        else if (beyond == 2) wait_vmcnt<4>();
        else if (beyond == 1) wait_vmcnt<2>();
        else wait_vmcnt<0>();
      }
      pp_fence();
      __builtin_amdgcn_s_barrier();
      // ---- MFMA section: one 64 x 32 quadrant over the whole K-tile, pinned between its two barriers
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      pp_fence();
      pp_prio<1>();
      const int ih = ph < 2 ? 0 : 1, jh = (ph == 0 || ph == 3) ? 0 : 1;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int bi = 0; bi < HA; ++bi)
#pragma unroll
          for (int bj = 0; bj < HW; ++bj) {
#ifdef LIST_PP_NO_MFMA       // ablation: operands stay live, no MFMA issued
            asm volatile("" ::"v"(a[bi][ks]), "v"(w[jh][bj][ks]));
#else
            auto& c = acc[ih * HA + bi][jh * HW + bj];
            if constexpr (X3 != 0) {
              if (ks == 0) {                 // [0] = hi halfs, [1] = lo halfs of the same 32 columns
                if constexpr (X3 == 3) {
                  c = mfma16<0>(a[bi][1], w[jh][bj][0], c);
                  c = mfma16<0>(a[bi][0], w[jh][bj][1], c);
                }
                c = mfma16<0>(a[bi][0], w[jh][bj][0], c);
              }
            } else if constexpr (S16) c = mfma16<FP16>(a[bi][ks], w[jh][bj][ks], c);
            else c = mfma<FP16>(a[bi][ks], w[jh][bj][ks], c);
#endif
          }
      pp_prio<0>();
      pp_fence();
      __builtin_amdgcn_s_barrier();
      pp_fence();
    }
  };
  int t = 0;
  for (; t + 3 <= nk; ++t) k_tile(t, std::true_type());       // tiles 0 .. nk-3
  for (; t < nk; ++t) k_tile(t, std::false_type());           // the last two: fewer quarters left to fly
  if (wm == 0) __builtin_amdgcn_s_barrier();     // the first group waits for the second to catch up
  if constexpr (S16) gemm_epilogue16<EPI, FP16>(p, acc, smem, m0, n0, wm, wn, wave, lane, dx_out);
  else gemm_epilogue<EPI, FP16>(p, acc, smem, m0, n0, wm, wn, wave, lane);
}

template <int TERMS, int EPI, int FP16>
__global__ __launch_bounds__(512, 2) void k_gemm_nt16(GemmParams p) {
  __shared__ __attribute__((aligned(16))) char smem[kLdsBytes];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int tiles_n = p.N / BN;
  const int ntiles = (p.M / BM) * tiles_n;
  const int tile = xcd_contiguous_block(blockIdx.x, ntiles);
  const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
  if (p.tile_gate && p.tile_gate[tile / tiles_n] == 0) return;      // gated re-run (exact border semantics): uniform exit

  f32x4v acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;

  using P = Pipe<TERMS>;
  const int frow = lane & 15, fq = lane >> 4;
  const int fswz = P::swz(frow);                       // tile row offsets are multiples of 16
  const int a_row_off = (wm * 128 + frow) * P::kRowBytes;
  const int w_row_off = (wn * 64 + frow) * P::kRowBytes;

  const int nk = p.K / P::BK;
#pragma unroll
  for (int s = 0; s < P::kAhead; ++s)
    if (s < nk) stage_tiles<TERMS>(p, smem + s * P::kStageBytes, m0, n0, s * P::kRowBytes, wave, lane);
  for (int t = 0; t < nk; ++t) {
    const int younger = min(P::kAhead - 1, nk - 1 - t);
    if (younger >= 2) wait_vmcnt<2 * P::kLoadsPerStage>();
    else if (younger == 1) wait_vmcnt<P::kLoadsPerStage>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (t + P::kAhead < nk)
      stage_tiles<TERMS>(p, smem + ((t + P::kAhead) % P::kStages) * P::kStageBytes, m0, n0,
                         (t + P::kAhead) * P::kRowBytes, wave, lane);
    const char* cur = smem + (t % P::kStages) * P::kStageBytes;
#pragma unroll
    for (int s2 = 0; s2 < P::BK / 32; ++s2) {
      const int coff = ((4 * s2 + fq) ^ fswz) << 4;
      bf16x8 ah[8], al[8], wh[4], wl[4];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        ah[i] = *(const bf16x8*)(cur + a_row_off + i * 16 * P::kRowBytes + coff);
        if (TERMS == 3)
          al[i] = *(const bf16x8*)(cur + P::kPlaneBytes + a_row_off + i * 16 * P::kRowBytes + coff);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        wh[j] = *(const bf16x8*)(cur + P::kWOff + w_row_off + j * 16 * P::kRowBytes + coff);
        if (TERMS == 3)
          wl[j] = *(const bf16x8*)(cur + P::kWOff + P::kPlaneBytes + w_row_off + j * 16 * P::kRowBytes + coff);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (TERMS == 3) {
            acc[i][j] = mfma16<0>(al[i], wh[j], acc[i][j]);
            acc[i][j] = mfma16<0>(ah[i], wl[j], acc[i][j]);
          }
          acc[i][j] = mfma16<FP16>(ah[i], wh[j], acc[i][j]);
        }
    }
  }

  gemm_epilogue16<EPI, FP16>(p, acc, smem, m0, n0, wm, wn, wave, lane);
}

// ---- fc_1 + fc_2 + fc_out in ONE launch (fp16 operands, inference: no activation is kept) --------------------------
// network/modules.py:277-281 for one 256-row tile per workgroup.  H2 = relu(fc_1) never leaves the registers: the
// products are taken TRANSPOSED (D^T[n][row] = sum_k W[n][k] H[row][k]: the weight fragment is the MFMA's A operand,
// the activation fragment its B operand), every wave owns 32 rows and ALL 256 columns (8 x 1 waves: 16 x 2 tiles of
// 16 x 16).  Which fc_1 column sits in which MFMA row is free -- it is only the row address of the W1 fragment -- and is
// chosen so that row m of column tiles 2 ks / 2 ks + 1 is column 32 ks + 8 (m >> 2) + (m & 3) (+ 4): a lane's accumulators
// of the two tiles (four columns each, bias + ReLU + fp16 in place) are then columns 32 ks + 8 q .. + 7 (q = lane >> 4),
// i.e. exactly its B fragment of fc_2's k-step ks in natural k order.  Same products, same k order, same summation tree
// in the fc_out epilogue as the two-launch path: bit-identical to it (asserted: a training forward, which keeps H1 / H2
// for the backward and takes the two launches, equals the inference forward bit for bit: tests/test_fused_tail_gpu.py, tests/test_threads_gpu.py).
// LDS: fc_1 streams H1 / W1 through the two 64-KB stages of the plain loop; W2 (128 KB) is resident for fc_2 -- its
// first K-tile prefetched into the last 32 KB at the start, the other three into the dead stages behind fc_1's last
// K-tile, waited for tile by tile (counted vmcnt).  What it removes against the two launches: H2's round trip (82 MB
// written and read per 160 k points), fc_1's LDS-staged store epilogue, fc_2's operand prologue, a queue boundary.
struct TailParams {
  GemmParams fc1;                   // a_hi = H1 [M][K], w_hi = W1 [256][K], bias = b1, K = H1 width, M
  const char* w2; const float* b2;  // [256][256] fp16, [256]
  const float* w3; const float* b3; float* sdf; const int* order; int n_valid;
};
constexpr int kTailLds = 163840;

__global__ __launch_bounds__(512, 2) void k_mlp_tail_f16(TailParams tp) {
  using P = Pipe<1>;
  __shared__ __attribute__((aligned(16))) char smem[kTailLds];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int m0 = xcd_contiguous_block(blockIdx.x, gridDim.x) * BM;
  const int col = lane & 15, q = lane >> 4;
  const int fswz = P::swz(col);                         // rows of a fragment are (multiple of 16) + col
  const GemmParams& p = tp.fc1;

  // W2 K-tile kt: [256 n][64 k] as 128-B rows, 16-B chunks XOR-swizzled like every operand image (on the source address)
  auto w2_base = [&](int kt) -> char* { return smem + (kt == 0 ? 131072 : (kt - 1) * 32768); };
  auto stage_w2 = [&](int kt) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int piece = 4 * wave + r;
      const int row = piece * 8 + lane / 8;
      const int chunk = (lane % 8) ^ P::swz(row);
      glds16(tp.w2 + (int64_t)row * 512 + kt * 128 + chunk * 16, w2_base(kt) + piece * 1024);
    }
  };
  stage_w2(0);

  // ---- fc_1: plain 2-stage loop (as k_gemm_nt16), 8 x 1 waves, transposed products
  f32x4v acc[16][2];
#pragma unroll
  for (int i = 0; i < 16; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
  const int h_row_off = (wave * 32 + col) * P::kRowBytes;
  // fc_1 column of MFMA row `col` in an even / odd column tile of a pair (see the header), and its chunk swizzle
  const int wrow_e = 8 * (col >> 2) + (col & 3), wrow_o = wrow_e + 4;
  const int wswz_e = P::swz(wrow_e), wswz_o = P::swz(wrow_o);
  const int nk = p.K / P::BK;
  stage_tiles<1>(p, smem, m0, 0, 0, wave, lane);
  for (int t = 0; t < nk; ++t) {
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (t + 1 < nk) stage_tiles<1>(p, smem + ((t + 1) & 1) * P::kStageBytes, m0, 0, (t + 1) * P::kRowBytes, wave, lane);
    const char* cur = smem + (t & 1) * P::kStageBytes;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const int coff = ((4 * s2 + q) ^ fswz) << 4;
      bf16x8 hf[2];
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) hf[rt] = *(const bf16x8*)(cur + h_row_off + rt * 16 * P::kRowBytes + coff);
#pragma unroll
      for (int nt = 0; nt < 16; ++nt) {
        // (32 (nt >> 1) is a multiple of 16: it does not change the swizzle of the row)
        const int wrow = 32 * (nt >> 1) + ((nt & 1) ? wrow_o : wrow_e);
        const bf16x8 wf = *(const bf16x8*)(cur + P::kWOff + wrow * P::kRowBytes + (((4 * s2 + q) ^ ((nt & 1) ? wswz_o : wswz_e)) << 4));
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) acc[nt][rt] = mfma16<1>(wf, hf[rt], acc[nt][rt]);
      }
    }
  }
  // bias of fc_1 for this lane's columns (32 (nt >> 1) + 8 q + 4 (nt & 1) + e), requested before the W2 stream
  float4 b1v[16];
#pragma unroll
  for (int nt = 0; nt < 16; ++nt) b1v[nt] = *(const float4*)(p.bias + 32 * (nt >> 1) + 8 * q + 4 * (nt & 1));
  __syncthreads();                                       // every wave is done with fc_1's stages (and b1 has landed)
  stage_w2(1); stage_w2(2); stage_w2(3);
  // H2 = relu(fc_1 + b1) in fp16 (the rounding of the two-launch path: saturating RNE), packed as fc_2's B fragments
  uint2 h2[16][2];
#pragma unroll
  for (int nt = 0; nt < 16; ++nt) {
    const float bb[4] = {b1v[nt].x, b1v[nt].y, b1v[nt].z, b1v[nt].w};
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = relu_nan(acc[nt][rt][e] + bb[e]);
      h2[nt][rt] = make_uint2(f2h2(v[0], v[1]), f2h2(v[2], v[3]));
    }
  }
  // ---- fc_2: K = 256 = 8 k-steps; accumulators D2^T[n2][row]
  f32x4v acc2[16][2];
#pragma unroll
  for (int i = 0; i < 16; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc2[i][j][e] = 0.f;
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    if (ks == 2) { wait_vmcnt<8>(); __builtin_amdgcn_s_barrier(); }      // W2 tile 1 has landed for everybody
    if (ks == 4) { wait_vmcnt<4>(); __builtin_amdgcn_s_barrier(); }
    if (ks == 6) { wait_vmcnt<0>(); __builtin_amdgcn_s_barrier(); }
    const char* wt = w2_base(ks >> 1);
    const int coff = ((4 * (ks & 1) + q) ^ fswz) << 4;
    bf16x8 hb[2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
      hb[rt] = __builtin_bit_cast(bf16x8, make_uint4(h2[2 * ks][rt].x, h2[2 * ks][rt].y, h2[2 * ks + 1][rt].x, h2[2 * ks + 1][rt].y));
#pragma unroll
    for (int mt = 0; mt < 16; ++mt) {
      const bf16x8 wf = *(const bf16x8*)(wt + (mt * 16 + col) * P::kRowBytes + coff);
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) acc2[mt][rt] = mfma16<1>(wf, hb[rt], acc2[mt][rt]);
    }
  }
  // ---- fc_out: sdf[row] = b3 + sum_n2 relu(fc_2 + b2)[n2] w3[n2] in the summation order of gemm_epilogue16's fused
  // epilogue (bit-identical to it): per 64-column group g, column c = 4 q + e of the group's four tiles summed in tile
  // order, then the butterfly over the 16 columns (xor 8, 4: the other lane groups; xor 2, 1: this lane's e), then
  // ((g0 + g1) + (g2 + g3)) + b3
  float grp[2][4];
#pragma unroll
  for (int gq = 0; gq < 4; ++gq) {
    float sc[2][4];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int e = 0; e < 4; ++e) sc[rt][e] = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int mt = 4 * gq + j;
      const float4 b2v = *(const float4*)(tp.b2 + 16 * mt + 4 * q), w3v = *(const float4*)(tp.w3 + 16 * mt + 4 * q);
      const float bb[4] = {b2v.x, b2v.y, b2v.z, b2v.w}, ww[4] = {w3v.x, w3v.y, w3v.z, w3v.w};
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int e = 0; e < 4; ++e) sc[rt][e] += relu_nan(acc2[mt][rt][e] + bb[e]) * ww[e];
    }
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        sc[rt][e] += __shfl_xor(sc[rt][e], 32);       // column c ^ 8  <->  lane group q ^ 2
        sc[rt][e] += __shfl_xor(sc[rt][e], 16);       // column c ^ 4  <->  lane group q ^ 1
      }
      const float a0 = sc[rt][0] + sc[rt][2], a1 = sc[rt][1] + sc[rt][3];     // column c ^ 2
      grp[rt][gq] = a0 + a1;                                                    // column c ^ 1
    }
  }
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
    const int row = m0 + wave * 32 + rt * 16 + col;
    if (q == 0 && row < tp.n_valid)
      tp.sdf[tp.order ? tp.order[row] : row] = ((grp[rt][0] + grp[rt][1]) + (grp[rt][2] + grp[rt][3])) + tp.b3[0];
  }
}

// fc1: the GemmParams of the fc_1 call (fp16; K % 64 == 0, N == 256); the rest: fc_2 / fc_out of the EPI_RELU_DOT call
hipError_t launch_mlp_tail(const GemmParams& fc1, const char* w2, const float* b2, const float* w3, const float* b3,
                           float* sdf, const int* order, int n_valid, hipStream_t s) {
  if (fc1.fmt != FMT_FP16 || fc1.N != BN || fc1.M % BM || fc1.K % 64 || fc1.K < 64 || !fc1.bias) return hipErrorInvalidValue;
  TailParams tp;
  tp.fc1 = fc1; tp.w2 = w2; tp.b2 = b2; tp.w3 = w3; tp.b3 = b3; tp.sdf = sdf; tp.order = order; tp.n_valid = n_valid;
  hipLaunchKernelGGL(k_mlp_tail_f16, dim3(fc1.M / BM), dim3(512), 0, s, tp);
  return hipGetLastError();
}

template <int TERMS, int EPI, int FP16>
static hipError_t launch_one(const GemmParams& p, hipStream_t s) {
  const int ntiles = (p.M / BM) * (p.N / BN);
  // interleaved split operands (fc_0's X and packed weight in the bf16 formats): always the ping-pong schedule
  if constexpr (FP16 == 0 && (EPI == EPI_RELU_SPLIT || EPI == EPI_F32 || EPI == EPI_DX)) {
    if (p.x3i) {
      if constexpr (TERMS == 3) hipLaunchKernelGGL((k_gemm_nt_pp<EPI, 0, true, 3>), dim3(ntiles), dim3(512), 0, s, p);
      else hipLaunchKernelGGL((k_gemm_nt_pp<EPI, 0, true, 1>), dim3(ntiles), dim3(512), 0, s, p);
      return hipGetLastError();
    }
  }
  if (p.x3i) return hipErrorInvalidValue;
  // operand strides, a row limit or a row vector exist in the ping-pong kernel only
  const bool need_pp = p.lda || p.ldw || p.a_rows || p.rowvec || p.k_gap || p.n_groups;
  // MFMA shape per epilogue (measured, fp16, P = 160k): 32x32x16 has the cheaper 64-B store runs on the short-K
  // layers (fc_1 0.074 vs 0.086 ms), 16x16x32 the cheaper row reduction of the fused fc_2 + fc_out epilogue (0.045
  // vs 0.058 ms) and, on the long-K ping-pong schedule, the higher clock (fc_0 0.55 -> 0.49 ms)
  if constexpr (EPI == EPI_RELU_DOT)
    hipLaunchKernelGGL((k_gemm_nt16<TERMS, EPI, FP16>), dim3(ntiles), dim3(512), 0, s, p);
  else {
#ifndef LIST_GEMM_NO_PINGPONG
    // the ping-pong schedule pays from 8 K-tiles on (fc_0: 57 K-tiles; fc_1 0.088 -> 0.081 ms and dX 0.82 -> 0.78 ms at
    // 8); at 4 K-tiles (dH) its prologue and stagger cancel the gain and the plain 2-stage loop stays.  plain_loop
    // (diagnostic) takes the plain loop of the SAME MFMA shape: bit-identical results, which makes it the schedule's
    // race detector.
#ifndef LIST_PP_MIN_K
#define LIST_PP_MIN_K 512
#endif
    if constexpr (TERMS == 1 && (EPI == EPI_RELU_SPLIT || EPI == EPI_F32 || (kPpShape16 && (EPI == EPI_DX || EPI == EPI_MASK_SPLIT)))) {
      if (p.K >= LIST_PP_MIN_K || need_pp) {
        if (!p.plain_loop) hipLaunchKernelGGL((k_gemm_nt_pp<EPI, FP16, kPpShape16>), dim3(ntiles), dim3(512), 0, s, p);
        else if (kPpShape16) hipLaunchKernelGGL((k_gemm_nt16<TERMS, EPI, FP16>), dim3(ntiles), dim3(512), 0, s, p);
        else hipLaunchKernelGGL((k_gemm_nt<TERMS, EPI, FP16>), dim3(ntiles), dim3(512), 0, s, p);
        return hipGetLastError();
      }
    }
#endif
    if (need_pp) return hipErrorInvalidValue;
#ifndef LIST_X3_SHAPE32      // the hi/lo-split long-K product (fc_0 in bf16x3) on the 16x16x32 shape: 1.52 -> 1.43 ms
    if constexpr (TERMS == 3 && (EPI == EPI_RELU_SPLIT || EPI == EPI_F32)) {
      if (p.K >= 1024) {
        hipLaunchKernelGGL((k_gemm_nt16<TERMS, EPI, FP16>), dim3(ntiles), dim3(512), 0, s, p);
        return hipGetLastError();
      }
    }
#endif
#ifndef LIST_BWD_SHAPE32     // the backward's data-gradient products (dX, dH) on the 16x16x32 shape: dX 0.87 -> 0.85 ms
    if constexpr (EPI == EPI_DX || EPI == EPI_MASK_SPLIT) {
      hipLaunchKernelGGL((k_gemm_nt16<TERMS, EPI, FP16>), dim3(ntiles), dim3(512), 0, s, p);
      return hipGetLastError();
    }
#endif
    hipLaunchKernelGGL((k_gemm_nt<TERMS, EPI, FP16>), dim3(ntiles), dim3(512), 0, s, p);
  }
  return hipGetLastError();
}

template <int TERMS, int FP16>
static hipError_t launch_epi(const GemmParams& p, int epi, hipStream_t s) {
  if (epi == EPI_RELU_SPLIT) return launch_one<TERMS, EPI_RELU_SPLIT, FP16>(p, s);
  if (epi == EPI_F32) return launch_one<TERMS, EPI_F32, FP16>(p, s);
  if (epi == EPI_MASK_SPLIT) return launch_one<TERMS, EPI_MASK_SPLIT, FP16>(p, s);
  if (epi == EPI_DX) return launch_one<TERMS, EPI_DX, FP16>(p, s);
  return launch_one<TERMS, EPI_RELU_DOT, FP16>(p, s);
}

hipError_t launch_gemm(const GemmParams& p, int terms, int epi, hipStream_t s) {
  if (p.n_groups) {            // grouped launch: every group on its own terms, together M row tiles
    if (epi != EPI_DX || p.n_groups < 0 || p.n_groups > kGemmMaxGroups) return hipErrorInvalidValue;
    int tiles = 0;
    for (int i = 0; i < p.n_groups; ++i) {
      const GemmGroup& g = p.grp[i];
      if (g.K % 64 || g.K <= 0 || g.m_tiles <= 0 || g.a_rows <= 0 || g.a_rows > g.m_tiles * BM || !g.a || !g.w || !g.out) return hipErrorInvalidValue;
      tiles += g.m_tiles;
    }
    if (tiles * BM != p.M || p.plain_loop || !kPpShape16) return hipErrorInvalidValue;   // (the 16x16x32 ping-pong kernel only)
  }
  if (p.M % BM || p.N % BN || p.K % 64 || p.M <= 0) return hipErrorInvalidValue;
  if (p.x3i && p.fmt == FMT_FP16) return hipErrorInvalidValue;
  if (epi == EPI_RELU_DOT && p.N != BN) return hipErrorInvalidValue;
  if (p.fmt == FMT_FP16) return launch_epi<1, 1>(p, epi, s);
  if (terms == 3) return launch_epi<3, 0>(p, epi, s);
  return launch_epi<1, 0>(p, epi, s);
}

}  // namespace list
