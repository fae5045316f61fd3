// fc_0 with the perceptual block of its A operand produced ON CHIP (round 4; north_star: "LDS staging of the sampled
// feature tiles").  network/modules.py:46-53 (bilinear sample of the 137^2 map) + :275-276 (concat, fc_0 + ReLU) for a
// tile of 128 query points and ALL 512 output columns per workgroup:
//
//   * K-tiles 0 .. img_C/64 - 1 (the perceptual block leads the gather order, list_common.h): every thread loads the 4
//     taps of its two (row, 8-channel) items straight from the prepared map into registers -- two K-tiles ahead of the
//     MFMAs that consume them --, interpolates in fp32 with the arithmetic of k_gather_img (gather_math.h: same products,
//     same order, same fp16 rounding) and writes the fp16 result into the A stage in LDS.  Those 1024 columns of X are
//     never written to or read from HBM (0.33 GB + 0.36 GB per 160 k points), and k_gather_img is not launched.
//   * the other K-tiles (voxel levels, xyz) come from X by LDS-DMA as in k_gemm_nt_pp.
//   * W streams from L2 by LDS-DMA, one 64-KB K-tile of all 512 rows per step (a row tile sees W once; with the
//     256 x 256 tile of k_gemm_nt_pp it is read by two workgroups per row tile, each with half of it).
//
// Tile: 8 waves as 2 (M) x 4 (N), wave = 64 rows x 128 columns = 4 x 8 tiles of v_mfma_f32_16x16x32_f16 (128
// accumulator registers).  LDS 160 KB: W 2 x 64 KB, A 2 x 16 KB; rows of 128 B, 16-B chunks XOR-swizzled with (row>>1)&7
// (on the LDS-DMA source address / the ds_write address, and on the ds_read_b128 address) as in gemm_kernels.hip.
// Same MFMA, same operand roles and the same k order per output element as k_gemm_nt_pp: bit-identical to the unfused
// path (tests/test_fused_fc0_gpu.py).  X3 = 3 / 1: the bf16 split formats (operand rows of 32 hi + 32 lo halfs per
// 32-column K-tile, list_common.h xi_off; fp32 maps; three MFMAs per operand pair in k_gemm_nt_pp's order, or hi . hi
// alone for plain bf16).  The NaN probe / exact redo of flagged 256-row tiles is unchanged: the fix-up
// kernel rewrites the whole X row of such tiles (perceptual block included) and the gated k_gemm_nt_pp re-runs on it.
#include "list_common.h"
#include "mfma_common.h"
#include "point_math.h"
#include "gather_math.h"

namespace list {

constexpr int kFusedLds = 163840;
constexpr int kFW = 65536, kFA = 16384;            // bytes per W / A stage
constexpr int kFAOff = 2 * kFW;                     // A stages behind the two W stages


__device__ __forceinline__ int fswz(int row) { return (row >> 1) & 7; }

// per-thread record of one row's projection (project(), point_math.h), packed: what the four tap loads need
// (valid: bit 0 = the row is a query point, bits 1..4 = Proj::dead, the taps outside the map)
struct RowProj { int64_t base; int dx, dy; float w00, w01, w10, w11; int valid; };
// the same per ROW of the tile, in LDS for the epilogue of the PROJ variant (48 B: three 16-B reads)
struct RowProjRec { int64_t base; int dx, dy; float w00, w01, w10, w11; int valid, pad_[3]; };
constexpr int kFProjRecOff = 81920;                 // behind the epilogue's staging tiles (8 waves x 32 x kStageLd floats)

// PROJ (fp16 operands; list_prep_img_proj, ListQueryArgs.img_proj): the map holds fp.kept sampled channels followed by
// N = 512 PROJECTED channels per pixel -- the low-resolution encoder levels already multiplied by their columns of
// fc_0.  K-tiles 0 .. kept/64 - 1 are produced on chip as before, the projected levels' K-tiles do not exist (gp.k_gap),
// and the epilogue adds the bilinear sample of the projected channels to the accumulators before bias and ReLU: in the
// staged layout a thread owns (row, 8 consecutive columns) = (point, 8 consecutive projected channels), one 16-B load
// per tap.  No 2-D gather kernel, no row-vector buffer.
template <int X3, bool PROJ = false>
__global__ __launch_bounds__(512, 2) void k_fc0_fused(FusedFc0Params fp) {
  static_assert(!PROJ || X3 == 0, "the projected form is built for fp16 operands");
  constexpr bool F16 = X3 == 0;
  constexpr int KT = F16 ? 64 : 32;                 // feature columns per K-tile (128 operand bytes per row)
  using M = MapT<F16 ? 1 : 0>;
  __shared__ __attribute__((aligned(16))) char smem[kFusedLds];
  const GemmParams& p = fp.gp;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int ntiles = p.M / 128;
  const int m0 = xcd_contiguous_block(blockIdx.x, ntiles) * 128;
  const int nk = p.K / KT;
  const int np = fp.n_produced;
  const int64_t lda = (int64_t)(p.lda ? p.lda : p.K) * (F16 ? 2 : 4), ldw = lda;      // bytes per operand row (hi + lo interleaved)
  // byte offset of K-tile t in the operand rows (tiles from k_gap_at on lie k_gap tiles further: gemm_kernels.hip)
  const int gap_at = p.k_gap_at, gap = p.k_gap;
  auto ktb = [&](int t) { return (t + (t >= gap_at ? gap : 0)) * 128; };

  // ---- the two items of this thread in a produced K-tile: rows r0, r0 + 64, chunk c (16 B of map: 8 halfs / 4 floats)
  const int pc = tid & 7, pr = tid >> 3;
  RowProj rp[2];
  if (np > 0 || PROJ) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const Pt pt = load_point(fp.g, m0 + pr + 64 * h);
      const Proj q = project(fp.trans_mat + pt.b * 12, pt.x, pt.y, pt.z, fp.ms, fp.Ct, fp.clamp_hi);
      rp[h].base = (int64_t)pt.b * fp.ms * fp.ms * fp.Ct + q.o00 + pc * M::V;
      rp[h].dx = q.o01 - q.o00; rp[h].dy = q.o10 - q.o00;        // (o11 = o00 + dx + dy: x1, y1 are clamped separately)
      rp[h].w00 = q.w00; rp[h].w01 = q.w01; rp[h].w10 = q.w10; rp[h].w11 = q.w11;
      rp[h].valid = (pt.valid ? 1 : 0) | (q.dead << 1);
    }
  }
  typename M::Raw taps[2][4];
  auto load_taps = [&](int t) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int64_t o = rp[h].base + t * KT;
      taps[h][0] = M::load(fp.img_map, o);
      taps[h][1] = M::load(fp.img_map, o + rp[h].dx);
      taps[h][2] = M::load(fp.img_map, o + rp[h].dy);
      taps[h][3] = M::load(fp.img_map, o + rp[h].dy + rp[h].dx);
    }
  };
  auto produce = [&](char* astage) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float r[M::V];
      tap_mul<M>(taps[h][0], rp[h].w00, r); tap_fma<M>(taps[h][1], rp[h].w01, r);
      tap_fma<M>(taps[h][2], rp[h].w10, r); tap_fma<M>(taps[h][3], rp[h].w11, r);
      const bool v = (rp[h].valid & 1) != 0;
      const int row = pr + 64 * h;
      if constexpr (F16) {
        const uint2 lo = half4_inrange(v ? make_float4(r[0], r[1], r[2], r[3]) : make_float4(0.f, 0.f, 0.f, 0.f));
        const uint2 hi = half4_inrange(v ? make_float4(r[4], r[5], r[6], r[7]) : make_float4(0.f, 0.f, 0.f, 0.f));
        *(uint4*)(astage + row * 128 + ((pc ^ fswz(row)) << 4)) = make_uint4(lo.x, lo.y, hi.x, hi.y);
      } else {
        // store_feat4<FMT_BF16_SPLIT> (list_common.h) into the LDS row: 4 hi halfs at column 4 pc, 4 lo halfs 32 columns on
        uint2 hi, lo;
        split4(v ? make_float4(r[0], r[1], r[2], r[3]) : make_float4(0.f, 0.f, 0.f, 0.f), hi, lo);
        char* base = astage + row * 128 + (pc & 1) * 8;
        *(uint2*)(base + (((pc >> 1) ^ fswz(row)) << 4)) = hi;
        *(uint2*)(base + (((4 + (pc >> 1)) ^ fswz(row)) << 4)) = lo;
      }
    }
  };

  // ---- LDS-DMA staging: W K-tile (64 pieces of 8 rows, 8 per wave), A K-tile from X (16 pieces, 2 per wave)
  const int a_last = p.M - 1;
  auto stage_w = [&](int t, char* wstage) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int piece = 8 * wave + r;
      const int row = piece * 8 + (lane >> 3);
      const int chunk = (lane & 7) ^ fswz(row);
      glds16(p.w_hi + (int64_t)row * ldw + ktb(t) + chunk * 16, wstage + piece * 1024);
    }
  };
  auto stage_a = [&](int t, char* astage) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int piece = 2 * wave + r;
      const int row = piece * 8 + (lane >> 3);
      const int chunk = (lane & 7) ^ fswz(row);
      // (non-temporal like k_gemm_nt_pp's A stream: default policy measured 0.640 -> 0.651 ms, round 4)
      glds16_nt(p.a_hi + (int64_t)min(m0 + row, a_last) * lda + ktb(t) + chunk * 16, astage + piece * 1024);
    }
  };

  f32x4v acc[4][8];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;

  const int frow = lane & 15, fo = lane >> 4;
  const int fsw = fswz(frow);                                   // block row offsets are multiples of 16
  const int a_row_off = (wm * 64 + frow) * 128;
  const int w_row_off = (wn * 128 + frow) * 128;

  // ---- prologue: tile 0 (produced or staged), taps of tile 1 in flight
  if (np > 0) {
    load_taps(0);
    wait_vmcnt<0>();
    produce(smem + kFAOff);
    stage_w(0, smem);
    if (np > 1) load_taps(1);
  } else {
    stage_w(0, smem);
    stage_a(0, smem + kFAOff);
  }

  for (int t = 0; t < nk; ++t) {
    // W(t) (and A(t) when it is staged) have landed; the taps of tile t + 1 (8 younger loads) may stay in flight.
    // My ds_writes of a produced A(t) are complete.
    if (t + 1 < np) wait_vmcnt<8>(); else wait_vmcnt<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const int nxt = (t + 1) & 1;
    const bool more = t + 1 < nk;
    const bool next_staged = more && (t + 1 >= np);
    if (more) stage_w(t + 1, smem + nxt * kFW);
    if (next_staged) stage_a(t + 1, smem + kFAOff + nxt * kFA);

    const char* aw = smem + kFAOff + (t & 1) * kFA;
    const char* ww = smem + (t & 1) * kFW;
    if constexpr (F16) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int coff = ((4 * ks + fo) ^ fsw) << 4;
        bf16x8 a[4], w[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = *(const bf16x8*)(aw + a_row_off + i * 16 * 128 + coff);
#pragma unroll
        for (int j = 0; j < 8; ++j) w[j] = *(const bf16x8*)(ww + w_row_off + j * 16 * 128 + coff);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[i][j] = mfma16<1>(a[i], w[j], acc[i][j]);
      }
    } else {
      // k-step 0 = the hi halfs, k-step 1 = the lo halfs of the same 32 columns; the products in k_gemm_nt_pp's order
      const int c_hi = (fo ^ fsw) << 4, c_lo = ((4 + fo) ^ fsw) << 4;
      bf16x8 ah[4], al[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ah[i] = *(const bf16x8*)(aw + a_row_off + i * 16 * 128 + c_hi);
        if (X3 == 3) al[i] = *(const bf16x8*)(aw + a_row_off + i * 16 * 128 + c_lo);
      }
#pragma unroll
      for (int jh = 0; jh < 2; ++jh) {
        bf16x8 wh[4], wl[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          wh[j] = *(const bf16x8*)(ww + w_row_off + (4 * jh + j) * 16 * 128 + c_hi);
          if (X3 == 3) wl[j] = *(const bf16x8*)(ww + w_row_off + (4 * jh + j) * 16 * 128 + c_lo);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            f32x4v c = acc[i][4 * jh + j];
            if constexpr (X3 == 3) {
              c = mfma16<0>(al[i], wh[j], c);
              c = mfma16<0>(ah[i], wl[j], c);
            }
            acc[i][4 * jh + j] = mfma16<0>(ah[i], wh[j], c);
          }
      }
    }

    if (t + 1 < np) {
      // the taps of tile t + 1 were requested one iteration ago, before this iteration's 8 W pieces.  (Measured and
      // dropped, round 4: the two waves of a SIMD taking this block on opposite sides of their MFMAs, 0.642 -> 0.669 ms.)
      wait_vmcnt<8>();
      produce(smem + kFAOff + nxt * kFA);
      if (t + 2 < np) load_taps(t + 2);
    }
  }

  // ---- epilogue: bias + ReLU + fp16, staged through LDS (32 rows x 64 columns per wave and turn) for 16-B stores
  {
    float* tile = (float*)smem + wave * (32 * kStageLd);
    const int col_in = lane & 15, row_in = 4 * (lane >> 4);
    const int rr = lane >> 3, c8 = (lane & 7) * 8;
    const int64_t row_base = m0 + wm * 64;
    const int col_base = wn * 128;
    bool bad = false;
    __syncthreads();                                 // every wave is done with the operand stages
    const RowProjRec* recs = (const RowProjRec*)(smem + kFProjRecOff);
    if constexpr (PROJ) {
      // the projections of the tile's 128 rows, from the threads that hold them (chunk 0 of rows pr and pr + 64)
      if (pc == 0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          RowProjRec rc;
          rc.base = rp[h].base; rc.dx = rp[h].dx; rc.dy = rp[h].dy;
          rc.w00 = rp[h].w00; rc.w01 = rp[h].w01; rc.w10 = rp[h].w10; rc.w11 = rp[h].w11; rc.valid = rp[h].valid;
          rc.pad_[0] = rc.pad_[1] = rc.pad_[2] = 0;
          *(RowProjRec*)(smem + kFProjRecOff + (pr + 64 * h) * (int)sizeof(RowProjRec)) = rc;
        }
      }
      __syncthreads();
    }
#pragma unroll
    for (int ih = 0; ih < 2; ++ih)
#pragma unroll
      for (int jh = 0; jh < 2; ++jh) {
        // PROJ: the four taps of this turn's four (row, 8 projected channels) items, requested before the staging.
        // (Round 4b, one box, two interleaved pairs: the sampling costs 0.07 ms of the kernel's 0.545 -- without the
        // loads 0.474 --; requesting turn t + 1's taps behind turn t's staging stores, two tap buffers, made it SLOWER,
        // 0.545 -> 0.563 ms with 108 instead of 68 B of scratch: not round-trip latency, the bytes themselves.)
        [[maybe_unused]] uint4 ptap[4][4];
#ifdef LIST_FUSED_PROJ_NO_SAMPLE        // ablation (wrong results): the epilogue without its tap loads
        if constexpr (false) {
#else
        if constexpr (PROJ) {
#endif
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const RowProjRec& rc = recs[wm * 64 + ih * 32 + rr + 8 * k];
            const int64_t o = rc.base + fp.kept + col_base + jh * 64 + c8;
            ptap[k][0] = M::load(fp.img_map, o);
            ptap[k][1] = M::load(fp.img_map, o + rc.dx);
            ptap[k][2] = M::load(fp.img_map, o + rc.dy);
            ptap[k][3] = M::load(fp.img_map, o + rc.dy + rc.dx);
          }
        }
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
          for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int e = 0; e < 4; ++e)
              tile[(ii * 16 + row_in + e) * kStageLd + jj * 16 + col_in] = acc[2 * ih + ii][4 * jh + jj][e];
        __syncthreads();
        const int cb = col_base + jh * 64 + c8;
        const float4 b0 = *(const float4*)(p.bias + cb), b1 = *(const float4*)(p.bias + cb + 4);
        const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int r = rr + 8 * k;
          const float4 x = *(const float4*)(tile + r * kStageLd + c8);
          const float4 y = *(const float4*)(tile + r * kStageLd + c8 + 4);
          const float v[8] = {x.x, x.y, x.z, x.w, y.x, y.y, y.z, y.w};
          float o[8];
          if constexpr (PROJ) {
            // the projected sample of (row, 8 channels): out-of-map taps masked like the reference (k_gather_img's
            // OUT32 form, reduce_proj_exact), rows beyond the query contribute nothing; (acc + sample) + bias, the K sum
            // first, as k_gemm_nt_pp's row-vector epilogue adds it
            const RowProjRec& rc = recs[wm * 64 + ih * 32 + r];
            const float wt[4] = {rc.w00, rc.w01, rc.w10, rc.w11};
            float sm[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) sm[e] = 0.f;
#ifndef LIST_FUSED_PROJ_NO_SAMPLE
#pragma unroll
#else
#pragma unroll
            for (int tp = 0; tp < 0; ++tp) (void)tp;
            if (false)
#endif
            for (int tp = 0; tp < 4; ++tp) {
              float f[8];
              M::unpack(ptap[k][tp], f);
              const bool dead = (rc.valid >> (1 + tp)) & 1;
#pragma unroll
              for (int e = 0; e < 8; ++e) sm[e] = fmaf(dead ? 0.f : f[e], wt[tp], sm[e]);
            }
            const bool live = (rc.valid & 1) != 0;
#pragma unroll
            for (int e = 0; e < 8; ++e) { o[e] = relu_nan((v[e] + (live ? sm[e] : 0.f)) + bb[e]); bad = bad || (o[e] != o[e]); }
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) { o[e] = relu_nan(v[e] + bb[e]); bad = bad || (o[e] != o[e]); }
          }
          store8_planes<F16 ? 1 : 0>(p.out_hi, F16 ? nullptr : p.out_lo, (row_base + ih * 32 + r) * p.ldo + cb, o);
        }
        __syncthreads();
      }
    // NaN probe for the exact redo of the 256-row tile's gathers (gemm_kernels.hip, gemm_epilogue16)
    if (p.nan_tiles && __builtin_amdgcn_ballot_w64(bad) != 0 && lane == 0) p.nan_tiles[m0 / 256] = 1;
  }
}

bool fused_fc0_eligible(const GemmParams& gp, int img_f16, int img_C) {
  // (img_C: the channels produced on chip -- all of the perceptual block, or the kept levels of list_prep_img_proj)
  if (gp.N != 512 || gp.M <= 0 || gp.M % 128 || gp.K % 64 || img_C < 0 || img_C % 64 || img_C > gp.K || !gp.bias ||
      gp.rowvec || gp.a_rows || gp.tile_gate) return false;
  if (gp.lda != gp.ldw || (gp.k_gap && !gp.lda)) return false;
  // fp16 operands pair with fp16 maps; the bf16 formats (interleaved hi / lo operands) with fp32 maps
  return gp.fmt == FMT_FP16 ? (img_f16 != 0 && !gp.x3i) : (img_f16 == 0 && gp.x3i != 0);
}

// terms: 3 = bf16x3, 1 = plain bf16 (split formats only)
hipError_t launch_fc0_fused(const FusedFc0Params& fp, int terms, hipStream_t s) {
  const dim3 grid(fp.gp.M / 128);
  if (fp.proj) {
    if (fp.gp.fmt != FMT_FP16 || fp.gp.N != 512) return hipErrorInvalidValue;
    hipLaunchKernelGGL((k_fc0_fused<0, true>), grid, dim3(512), 0, s, fp);
  } else if (fp.gp.fmt == FMT_FP16) hipLaunchKernelGGL((k_fc0_fused<0, false>), grid, dim3(512), 0, s, fp);
  else if (terms == 3) hipLaunchKernelGGL((k_fc0_fused<3, false>), grid, dim3(512), 0, s, fp);
  else hipLaunchKernelGGL((k_fc0_fused<1, false>), grid, dim3(512), 0, s, fp);
  return hipGetLastError();
}

}  // namespace list
