// Adjoint of the matrix-core gather of a coarse voxel level (gather_box_kernels.hip; the reference's autograd of
// network/modules.py:256-265 for the 16^3 and 8^3 x 128-channel levels), fp16 operands: the gradient of the voxel box that a run
// of Morton-consecutive points touches is ONE small dense product on the matrix cores,
//   dV^T[c][v] = sum_k dX^T[c][k] * Wt[k][v],   k = (point, stencil slot j = 0..6 and a zero slot),  v = box row,
//   v_mfma_f32_16x16x32_f16: one K-step = 4 points x 8 slots,
// instead of the read-add-write chain of k_scatter_vox_win (bwd_scatter_kernels.hip) that sums the same terms on the
// VALU, one point and one window slot at a time.  Wt is 94 % zeros (8 taps of <= 128 box rows) and that does not matter:
// the dense product of a workgroup is 15 - 30 MFLOP, a few thousand matrix-core cycles.
//   A = dX^T: the run's dX rows are copied to LDS as they lie in memory ([sample][128 halfs], 16-B chunks XOR-swizzled
//       by the stencil slot, 8-B halfs swapped for odd points) and read with ds_read_b64_tr_b16, like the forward's V^T;
//   B = Wt:   formed per lane in the B-fragment layout -- a lane owns ONE box row (its voxel) and the 8 slots of ONE point
//       per K-step, so its 8 values are products of three per-axis factors looked up by comparing the voxel's coordinates
//       with the point's nine axis records; split hi + lo in fp16 so that the products are exact in the fp32 accumulator
//       (same interpolation arithmetic as the VALU kernels up to the order of the fp32 sums);
//   D:        wave w accumulates box-row tiles w and w + 4 (16 rows each) x all 128 channels, 64 accumulator registers.
// The finished box goes through LDS once more ([row][128 halfs] at the window scale kWinPkScale) so that the flush is the
// one k_scatter_vox_win has: packed-half atomics, lanes over channel pairs, whole 256-B rows per instruction.
//
// Workgroup = 64 consecutive rows (Morton order), 256 threads; runs = the forward's aligned power-of-two runs whose box
// has at most 128 rows (a single point's 4 x 4 x 4 always fits; at 16^3 that is ~8 points per run).
//
// Measured (config 2, 160 000 points, in-line backward, rocprofv3 kernel trace): 16^3 level 0.519 -> 0.224 ms, 8^3 level
// 0.28 -> 0.118 ms against k_scatter_vox_win; of the 0.239 ms it took before its row -> voxel divisions became multiply-shifts
// (0.224 now) the flush was 0.085 (exposed), the K loop 0.114 -- the weights
// on the VALU more than the 32 MFMAs of a K-step: hi-only weights save 0.02; the per-axis factors read from per-point
// LDS tables instead of compared and selected per lane: 0.224 -> 0.231 ms, dropped --, and 0.08 is what reading 287 MB of
// dX, the point records and the partition take (profiles/r04b_box_adjoint.txt).  fp16 training step: 6.63 -> 6.29 ms.
#include "list_common.h"
#include "point_math.h"
#include "box_partition.h"

namespace list {

constexpr int kAdjPts = 64;                       // points per workgroup
constexpr int kAdjRows = 128;                     // box rows (8 tiles of 16: two per wave)
constexpr int kAdjChunkPts = 8;                   // points staged per chunk (two K-steps)
constexpr int kAdjC = 128;
constexpr float kAdjPkScale = 0.0625f;            // = kWinPkScale of bwd_scatter_kernels.hip (asserted by the launcher)

struct AdjLds {
  static constexpr int kRowBytes = 2 * kAdjC;                                   // 256
  static constexpr int kStageRows = kAdjChunkPts * LIST_N_STENCIL;              // 56
  static constexpr int stage = 0;                                               // 2 x [56][256 B]; later the box [128][256 B]
  static constexpr int stage_bytes = kStageRows * kRowBytes;                    // 14336
  static constexpr int region = kAdjRows * kRowBytes;                           // 32768 >= 2 * stage_bytes
  static constexpr int zero = region;                                           // one row of zeros
  static constexpr int ptab = zero + kRowBytes;                                 // AxisW [64][3 axes][3 variants]
  static constexpr int run = ptab + kAdjPts * 9 * (int)sizeof(AxisW);           // RunBox [64]
  static constexpr int pbox = run + kAdjPts * (int)sizeof(RunBox);              // int [64][4]
  static constexpr int total = pbox + kAdjPts * 16;
};
static_assert(2 * AdjLds::stage_bytes <= AdjLds::region, "the two staging buffers share the box's LDS");

// grid = rows / 64, block = 256.  img16: the level's zeroed fp16 image (gradient scale x kAdjPkScale).
// F32OUT (diagnostic, LIST_SCATTER_F32=1): the accumulators go straight to the level's zeroed fp32 gradient as float
// atomics, unrounded -- the form tests/test_box_adjoint_gpu.py compares with the window kernel's fp32 flush at 2e-5.
// (243 registers, two workgroups per CU; held to 168 for three the compiler spills 300 B and the kernel takes 2.2x as long)
template <int F32OUT>
__global__ __launch_bounds__(256, 2) void k_scatter_vox_box(ScatterParams sp, ListVoxLevel gv, int col_off,
                                                          _Float16* __restrict__ img16) {
  using L = AdjLds;
  constexpr int RB = L::kRowBytes;
  constexpr int NT = kAdjC / 16;                                // 16-channel tiles
  __shared__ __attribute__((aligned(16))) char smem[L::total];
  AxisW* ptab = (AxisW*)(smem + L::ptab);
  RunBox* runs = (RunBox*)(smem + L::run);
  int* pbox = (int*)(smem + L::pbox);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = uni(tid >> 6);
  const int blk = xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int64_t row0 = (int64_t)blk * kAdjPts;
  const int W = gv.W, H = gv.H, D = gv.D;

  // ---- 1a. waves 0..2: axis `wave` of the 64 points (as the forward); wave 3: the zero row ---------------------------
  if (wave < 3) {
    const Pt p = load_point(sp.g, (int)row0 + lane);
    const float c = wave == 0 ? p.x : (wave == 1 ? p.y : p.z);
    const int S = wave == 0 ? W : (wave == 1 ? H : D);
    const Axis a[3] = {axis_setup(c, S), axis_setup(c - kDisp, S), axis_setup(c + kDisp, S)};
#pragma unroll
    for (int v = 0; v < 3; ++v) {
      AxisW e;
      e.i0 = a[v].i0;
      e.w0 = p.valid ? a[v].w0 : 0.f;
      e.w1 = (p.valid && a[v].has1) ? a[v].w1 : 0.f;
      ptab[(lane * 3 + wave) * 3 + v] = e;
    }
    pbox[lane * 4 + wave] = a[1].i0 | ((a[2].i0 + a[2].has1) << 8);
    if (wave == 0) pbox[lane * 4 + 3] = p.valid ? p.b : -1;
  } else {
    *(unsigned*)(smem + L::zero + lane * 4) = 0u;
  }
  __syncthreads();
  // ---- 1b. wave 0: aligned power-of-two runs whose box fits -------------------------------------------------------------
  if (wave == 0) {
    const int4 pb = *(const int4*)(pbox + lane * 4);
    const bool valid = pb.w >= 0;
    SegBox sb;
    sb.f0 = valid ? (unsigned)((pb.x & 255) | ((pb.y & 255) << 16)) : 0x7fff7fffu;
    sb.f1 = valid ? (unsigned)((pb.z & 255) | ((255 - (pb.x >> 8)) << 16)) : 0x7fff7fffu;
    sb.f2 = valid ? (unsigned)((255 - (pb.y >> 8)) | ((255 - (pb.z >> 8)) << 16)) : 0x7fff7fffu;
    sb.bmin = valid ? pb.w : INT_MAX;
    sb.nbmax = valid ? ~pb.w : INT_MAX;
    int level = 0;
    SegBox best = sb;
#define LIST_SEG_STAGE(S)                                              \
    seg_merge<S>(sb);                                                  \
    if (level == S && seg_fits(sb, kAdjRows, INT_MAX)) { level = S + 1; best = sb; }
    LIST_SEG_STAGE(0) LIST_SEG_STAGE(1) LIST_SEG_STAGE(2) LIST_SEG_STAGE(3) LIST_SEG_STAGE(4) LIST_SEG_STAGE(5)
#undef LIST_SEG_STAGE
    if ((lane & ((1 << level) - 1)) == 0) {
      const bool any = best.bmin != INT_MAX;
      const int lox = best.f0 & 0xffff, loy = best.f0 >> 16, loz = best.f1 & 0xffff;
      const int hix = 255 - (int)(best.f1 >> 16), hiy = 255 - (int)(best.f2 & 0xffff), hiz = 255 - (int)(best.f2 >> 16);
      RunBox rb;
      rb.count = 1 << level;
      rb.b = any ? best.bmin : 0;
      rb.lo = any ? (lox | (loy << 8) | (loz << 16)) : 0;
      rb.n = any ? ((hix - lox + 1) | ((hiy - loy + 1) << 8) | ((hiz - loz + 1) << 16)) : 0;
      runs[lane] = rb;
    }
  }
  __syncthreads();

  const unsigned short* __restrict__ dx = (const unsigned short*)sp.dx;
  const int q = lane >> 4, col = lane & 15;                    // MFMA lane roles: point of the K-step / box row of the tile
  const int tr_r = (lane >> 2) & 3, tr_p = lane & 3;           // transposed read: slot within the 4-slot block, 4-channel group
  // byte offset of channel tile t in a staged row, as this lane reads it: tile t = channels 32 (t >> 1) + 8 p + 4 (t & 1)
  // + 0..3 for group p (the forward's assignment: a lane's accumulators of tiles 2u, 2u + 1 are 8 consecutive channels);
  // physical position: 16-B chunk ^ (slot & 3) << 2, 8-B halfs swapped for odd points (q & 1: a K-step starts at an even
  // point) -- the 32 lanes of a transposed read (2 points x 4 slots x 4 channel groups) hit 32 distinct 8-B bank slots
  int aoff[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
    aoff[t] = ((tr_p << 4) + (((t >> 1) << 6) | ((t & 1) << 3))) ^ ((tr_r << 6) | ((q & 1) << 3));

  int first = 0;
#pragma unroll 1
  while (first < kAdjPts) {
    const RunBox rb = runs[first];
    const int count = uni(rb.count), rb_b = uni(rb.b), rlo = uni(rb.lo), rn = uni(rb.n);
    const int lox = rlo & 255, loy = (rlo >> 8) & 255, loz = rlo >> 16;
    const int nx = rn & 255, ny = (rn >> 8) & 255, nz = rn >> 16;
    const int rows = nx * ny * nz;
    if (rows == 0) { first += count; continue; }                // no valid point in the run (uniform)
    const int n_vt = (rows + 15) >> 4;                          // box-row tiles in use
    const int inv_nx = (65536 + nx - 1) / nx, inv_ny = (65536 + ny - 1) / ny;
    // this lane's box rows (tiles wave, wave + 4) as absolute voxel coordinates; a row beyond the box matches nothing
    int vx[2], vy[2], vz[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int v = 16 * (wave + 4 * i) + col;
      const int yz = (v * inv_nx) >> 16, ix = v - yz * nx;
      const int iz = (yz * inv_ny) >> 16, iy = yz - iz * ny;
      const bool in = v < rows;
      vx[i] = in ? lox + ix : -4; vy[i] = in ? loy + iy : -4; vz[i] = in ? loz + iz : -4;
    }
    const bool own0 = wave < n_vt, own1 = wave + 4 < n_vt;      // (uniform)
    f32x4v acc[2][NT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[i][t] = (f32x4v){0.f, 0.f, 0.f, 0.f};

    // staging of chunk ci: rows (point, slot j < 7) x 16 chunks of 16 B = 896 pieces, 3.5 per thread
    const int nchunks = (count + kAdjChunkPts - 1) / kAdjChunkPts;
    uint4 sv[4];
    auto stage_load = [&](int ci) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = tid + 256 * e;
        const int r = i >> 4, chunk = i & 15;
        const int pl = r / LIST_N_STENCIL, j = r - pl * LIST_N_STENCIL;
        sv[e] = make_uint4(0u, 0u, 0u, 0u);
        if (i < L::kStageRows * 16 && ci * kAdjChunkPts + pl < count)
          sv[e] = *(const uint4*)(dx + (row0 + first + ci * kAdjChunkPts + pl) * sp.g.Kp + col_off + j * kAdjC + chunk * 8);
      }
    };
    auto stage_store = [&](int ci) {
      char* buf = smem + L::stage + (ci & 1) * L::stage_bytes;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = tid + 256 * e;
        if (i >= L::kStageRows * 16) continue;
        const int r = i >> 4, chunk = i & 15;
        const int pl = r / LIST_N_STENCIL, j = r - pl * LIST_N_STENCIL;
        const uint4 w = (pl & 1) ? make_uint4(sv[e].z, sv[e].w, sv[e].x, sv[e].y) : sv[e];
        *(uint4*)(buf + r * RB + ((chunk ^ ((j & 3) << 2)) << 4)) = w;
      }
    };
    stage_load(0);
#pragma unroll 1
    for (int ci = 0; ci < nchunks; ++ci) {
      stage_store(ci);
      __syncthreads();                 // (one barrier per chunk: the other buffer was last read before the previous barrier)
      if (ci + 1 < nchunks) stage_load(ci + 1);
      const char* buf = smem + L::stage + (ci & 1) * L::stage_bytes;
#ifdef LIST_ADJ_NO_MFMA            // ablation (wrong results; staging and barriers only): 0.239 -> 0.125 ms, 0.118 -> 0.070
      if (sp.g.Kp < 0)
#endif
#pragma unroll 1
      for (int ks = 0; ks < kAdjChunkPts / 4; ++ks) {
        const int plc = 4 * ks + q;                              // point of the chunk
        const int pl = ci * kAdjChunkPts + plc;                  // point of the run
        if (uni(ci * kAdjChunkPts + 4 * ks) >= count) break;     // (runs of 1, 2, 4 points: uniform)
        const bool live = pl < count;
        const int pt = first + (live ? pl : 0);
        // A: slots 0..3 and 4..7 of this lane's point; slot 7 and points beyond the run read the zero row
        const char* a_lo = live ? buf + (plc * LIST_N_STENCIL + tr_r) * RB : smem + L::zero;
        const char* a_hi = (live && tr_r < 3) ? buf + (plc * LIST_N_STENCIL + 4 + tr_r) * RB : smem + L::zero;
        s16x4 a0[NT], a1[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          a0[t] = tr_read16(a_lo + aoff[t]);
          a1[t] = tr_read16(a_hi + aoff[t]);
        }
        // B: the 7 weights of (point, slot) at this lane's voxel, per owned tile (the factor of an axis is the record's
        // w0 where the voxel is the base tap, w1 where it is the next one, 0 elsewhere)
        const AxisW* rec = ptab + pt * 9;
        float hx[2][3], hy[2][3], hz[2][3];
#pragma unroll
        for (int v = 0; v < 3; ++v) {
          const AxisW fx = rec[v], fy = rec[3 + v], fz = rec[6 + v];
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const int dxv = vx[i] - fx.i0, dyv = vy[i] - fy.i0, dzv = vz[i] - fz.i0;
            hx[i][v] = dxv == 0 ? fx.w0 : (dxv == 1 ? fx.w1 : 0.f);
            hy[i][v] = dyv == 0 ? fy.w0 : (dyv == 1 ? fy.w1 : 0.f);
            hz[i][v] = dzv == 0 ? fz.w0 : (dzv == 1 ? fz.w1 : 0.f);
          }
        }
        f16x8 bhi[2], blo[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          if (i == 0 ? !own0 : !own1) continue;
          const float cx = live ? hx[i][0] : 0.f;                  // (every slot's weight has a centre factor)
          const float yz = hy[i][0] * hz[i][0], xz = cx * hz[i][0], xy = cx * hy[i][0];
          const float w[8] = {cx * yz, hx[i][1] * yz, hx[i][2] * yz, hy[i][1] * xz, hy[i][2] * xz, hz[i][1] * xy, hz[i][2] * xy, 0.f};
          unsigned hi[4], lo[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            hi[e] = pk_h2(w[2 * e], w[2 * e + 1]);
            lo[e] = pk_h2(w[2 * e] - h2f((unsigned short)(hi[e] & 0xffffu)), w[2 * e + 1] - h2f((unsigned short)(hi[e] >> 16)));
          }
          bhi[i] = __builtin_bit_cast(f16x8, make_uint4(hi[0], hi[1], hi[2], hi[3]));
          blo[i] = __builtin_bit_cast(f16x8, make_uint4(lo[0], lo[1], lo[2], lo[3]));
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          if (i == 0 ? !own0 : !own1) continue;
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const f16x8 a = __builtin_bit_cast(f16x8, (s16x8){a0[t][0], a0[t][1], a0[t][2], a0[t][3], a1[t][0], a1[t][1], a1[t][2], a1[t][3]});
            acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bhi[i], acc[i][t], 0, 0, 0);
          }
#ifndef LIST_ADJ_HI_ONLY            // ablation (weights rounded to fp16): 16^3 level 0.253 -> 0.230 ms -- not worth the exactness
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const f16x8 a = __builtin_bit_cast(f16x8, (s16x8){a0[t][0], a0[t][1], a0[t][2], a0[t][3], a1[t][0], a1[t][1], a1[t][2], a1[t][3]});
            acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, blo[i], acc[i][t], 0, 0, 0);
          }
#endif
        }
      }
    }
    if (F32OUT) {
      // D: column = box row (lane & 15), rows 4 q + reg of tile t = channels 32 (t >> 1) + 8 q + 4 (t & 1) + reg
      const float inv_s = sp.scale[1];
      float* base = (float*)gv.data + (int64_t)rb_b * gv.image_stride;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (i == 0 ? !own0 : !own1) continue;
        const int v = 16 * (wave + 4 * i) + col;
        if (v >= rows) continue;
        const int yz = (v * inv_nx) >> 16, ix = v - yz * nx;
        const int iz = (yz * inv_ny) >> 16, iy = yz - iz * ny;
        float* dst = base + ((int64_t)((loz + iz) * H + (loy + iy)) * W + (lox + ix)) * kAdjC;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float val = acc[i][t][e];
            if (val != 0.f) atomicAdd(dst + 32 * (t >> 1) + 8 * q + 4 * (t & 1) + e, val * inv_s);
          }
      }
      first += count;
      if (first < kAdjPts) __syncthreads();                     // the next run stages over the buffers
      continue;
    }
    __syncthreads();                   // every wave is done with the staging buffers: the box takes their place
    // D: column = box row (lane & 15), rows 4 q + reg of tile t = channels 32 (t >> 1) + 8 q + 4 (t & 1) + reg -> 16-B
    // pieces of 8 consecutive channels, chunk XOR (row & 15) (the 16 rows of a tile land on distinct banks)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (i == 0 ? !own0 : !own1) continue;
      const int v = 16 * (wave + 4 * i) + col;
#pragma unroll
      for (int u = 0; u < NT / 2; ++u) {
        const f32x4v c0 = acc[i][2 * u], c1 = acc[i][2 * u + 1];
        const uint2 lo = half4_inrange(make_float4(c0[0] * kAdjPkScale, c0[1] * kAdjPkScale, c0[2] * kAdjPkScale, c0[3] * kAdjPkScale));
        const uint2 hi = half4_inrange(make_float4(c1[0] * kAdjPkScale, c1[1] * kAdjPkScale, c1[2] * kAdjPkScale, c1[3] * kAdjPkScale));
        *(uint4*)(smem + L::stage + v * RB + (((4 * u + q) ^ col) << 4)) = make_uint4(lo.x, lo.y, hi.x, hi.y);
      }
    }
    __syncthreads();
    // flush: lanes over channel pairs, one box row per wave and pass -- 256 contiguous bytes per atomic instruction
#ifdef LIST_ADJ_NO_FLUSH           // ablation (wrong results): 16^3 level 0.239 -> 0.154 ms, 8^3 0.118 -> 0.104
    if (sp.g.Kp < 0)
#endif
    {
      typedef _Float16 half2v __attribute__((ext_vector_type(2)));
      _Float16* base16 = img16 + (int64_t)rb_b * gv.image_stride;
      // one box row per wave and pass (its voxel address is scalar arithmetic), lane = channel pair 2 lane, 2 lane + 1
#pragma unroll 1
      for (int v = wave; v < rows; v += 4) {
        const int yz = (v * inv_nx) >> 16, ix = v - yz * nx;           // (exact for v < 256: the forward's box copy)
        const int iz = (yz * inv_ny) >> 16, iy = yz - iz * ny;
        const unsigned bits = *(const unsigned*)(smem + L::stage + v * RB + ((((lane >> 2) ^ (v & 15)) << 4) | ((lane & 3) << 2)));
        if ((bits & 0x7fff7fffu) == 0u) continue;
        __builtin_amdgcn_global_atomic_fadd_v2f16(
            (__attribute__((address_space(1))) half2v*)(base16 + ((int64_t)((loz + iz) * H + (loy + iy)) * W + (lox + ix)) * kAdjC + 2 * lane),
            __builtin_bit_cast(half2v, bits));
      }
    }
    first += count;
    if (first < kAdjPts) __syncthreads();                       // the next run stages over the box
  }
}

// a window level (stencil shorter than a voxel), fp16 dX, 128 channels, image scaled by `pk_scale`
bool scatter_f32_diagnostic() {
  static const bool on = [] { const char* e = getenv("LIST_SCATTER_F32"); return e && e[0] == '1'; }();
  return on;
}

bool scatter_box_eligible(const ScatterParams& sp, const ListVoxLevel& gv, int col_off, float pk_scale) {
  static const bool off = [] { const char* e = getenv("LIST_SCATTER_BOX"); return e && e[0] == '0' && e[1] == 0; }();
  if (off || !sp.dx_f16 || gv.C != kAdjC || pk_scale != kAdjPkScale) return false;
  if ((col_off % 8) != 0 || (sp.g.Kp % 8) != 0 || (gv.image_stride % 2) != 0) return false;
  if (gv.W > 255 || gv.H > 255 || gv.D > 255) return false;   // 8-bit coordinates in the run records
  return (sp.g.rows % kAdjPts) == 0;
}

hipError_t launch_scatter_vox_box(const ScatterParams& sp, const ListVoxLevel& gv, int col_off, _Float16* img16,
                                  hipStream_t s) {
  if (img16) hipLaunchKernelGGL(k_scatter_vox_box<0>, dim3((unsigned)(sp.g.rows / kAdjPts)), dim3(256), 0, s, sp, gv, col_off, img16);
  else hipLaunchKernelGGL(k_scatter_vox_box<1>, dim3((unsigned)(sp.g.rows / kAdjPts)), dim3(256), 0, s, sp, gv, col_off, img16);
  return hipGetLastError();
}

}  // namespace list
