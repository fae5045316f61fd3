// The matrix-core adjoint of the coarse voxel levels (bwd_box_kernels.hip) for the formats whose dX is fp32 (bf16x3, bf16):
// the same product dV^T[c][v] = sum_k dX^T[c][k] * Wt[k][v] per run of Morton-consecutive points, with BOTH operands split
// into bf16 hi + lo planes and three MFMAs per product (hi * hi + hi * lo + lo * hi on v_mfma_f32_16x16x32_bf16: 16 mantissa
// bits per operand, the grade of the forward's bf16x3 products), the box accumulated in fp32 and flushed as float atomics
// into the level's zeroed gradient -- 256 contiguous bytes per instruction, like k_scatter_vox_win's fp32 flush.
//   staging: the dX rows of 8 points are loaded as fp32, split, and written as two [sample][128] bf16 images (the fp16
//            kernel's swizzle), the next chunk's fp32 pieces requested a chunk ahead (32 registers);
//   LDS:     64 KB for the fp32 box (the 2 x 2 staging images share it) + 8.5 KB of tables: two workgroups per CU.
// Runs, tiles, lane roles and the weights' arithmetic are those of k_scatter_vox_box.
#include "list_common.h"
#include "point_math.h"
#include "box_partition.h"
#include "mfma_common.h"

namespace list {

constexpr int kSplPts = 64;                       // points per workgroup
constexpr int kSplRows = 128;                     // box rows (8 tiles of 16: two per wave)
constexpr int kSplChunkPts = 8;                   // points staged per chunk (two K-steps)
constexpr int kSplC = 128;

struct SplLds {
  static constexpr int kRowBytes = 2 * kSplC;                                   // 256: one bf16 plane of a staged row
  static constexpr int kStageRows = kSplChunkPts * LIST_N_STENCIL;              // 56
  static constexpr int plane_bytes = kStageRows * kRowBytes;                    // 14336
  static constexpr int stage = 0;                                               // 2 buffers x (hi | lo); later the fp32 box
  static constexpr int region = kSplRows * kSplC * 4;                           // 65536 >= 4 * plane_bytes
  static constexpr int zero = region;                                           // one row of zeros
  static constexpr int ptab = zero + kRowBytes;                                 // AxisW [64][3 axes][3 variants]
  static constexpr int run = ptab + kSplPts * 9 * (int)sizeof(AxisW);           // RunBox [64]
  static constexpr int pbox = run + kSplPts * (int)sizeof(RunBox);              // int [64][4]
  static constexpr int total = pbox + kSplPts * 16;
};
static_assert(4 * SplLds::plane_bytes <= SplLds::region, "the staging images share the box's LDS");

__device__ __forceinline__ unsigned pk_bf2(float a, float b) { return (unsigned)f2bf(a) | ((unsigned)f2bf(b) << 16); }

// grid = rows / 64, block = 256; gv.data: the level's zeroed fp32 gradient
__global__ __launch_bounds__(256, 2) void k_scatter_vox_box_split(ScatterParams sp, ListVoxLevel gv, int col_off) {
  using L = SplLds;
  constexpr int RB = L::kRowBytes;
  constexpr int NT = kSplC / 16;                                // 16-channel tiles
  __shared__ __attribute__((aligned(16))) char smem[L::total];
  AxisW* ptab = (AxisW*)(smem + L::ptab);
  RunBox* runs = (RunBox*)(smem + L::run);
  int* pbox = (int*)(smem + L::pbox);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = uni(tid >> 6);
  const int blk = xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int64_t row0 = (int64_t)blk * kSplPts;
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
    if (level == S && seg_fits(sb, kSplRows, INT_MAX)) { level = S + 1; best = sb; }
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

  const float* __restrict__ dx = (const float*)sp.dx;
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

  const float inv_s = sp.scale[1];
  int first = 0;
#pragma unroll 1
  while (first < kSplPts) {
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

    // staging of chunk ci: rows (point, slot j < 7) x 16 pieces of 8 channels = 896 pieces, 3.5 per thread: 32 B of fp32
    // in, 16 B of hi and 16 B of lo out (chunk ^ (j & 3) << 2, 8-B halfs swapped for odd points: the fp16 kernel's image)
    const int nchunks = (count + kSplChunkPts - 1) / kSplChunkPts;
    float4 sva[4], svb[4];                                      // the next chunk's fp32 pieces, requested a chunk ahead
    auto stage_load = [&](int ci) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = tid + 256 * e;
        const int r = i >> 4, chunk = i & 15;
        const int pl = r / LIST_N_STENCIL, j = r - pl * LIST_N_STENCIL;
        sva[e] = make_float4(0.f, 0.f, 0.f, 0.f); svb[e] = sva[e];
        if (i < L::kStageRows * 16 && ci * kSplChunkPts + pl < count) {
          const float* src = dx + (row0 + first + ci * kSplChunkPts + pl) * sp.g.Kp + col_off + j * kSplC + chunk * 8;
          sva[e] = *(const float4*)src; svb[e] = *(const float4*)(src + 4);
        }
      }
    };
    auto stage_store = [&](int ci) {
      char* buf = smem + L::stage + (ci & 1) * (2 * L::plane_bytes);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = tid + 256 * e;
        if (i >= L::kStageRows * 16) continue;
        const int r = i >> 4, chunk = i & 15;
        const int pl = r / LIST_N_STENCIL, j = r - pl * LIST_N_STENCIL;
        uint2 h0, h1, l0, l1;
        split4(sva[e], h0, l0);
        split4(svb[e], h1, l1);
        const int off = r * RB + ((chunk ^ ((j & 3) << 2)) << 4);
        const bool odd = pl & 1;
        *(uint4*)(buf + off) = odd ? make_uint4(h1.x, h1.y, h0.x, h0.y) : make_uint4(h0.x, h0.y, h1.x, h1.y);
        *(uint4*)(buf + L::plane_bytes + off) = odd ? make_uint4(l1.x, l1.y, l0.x, l0.y) : make_uint4(l0.x, l0.y, l1.x, l1.y);
      }
    };
    stage_load(0);
#pragma unroll 1
    for (int ci = 0; ci < nchunks; ++ci) {
      stage_store(ci);
      __syncthreads();                 // (one barrier per chunk: the other buffer was last read before the previous barrier)
      if (ci + 1 < nchunks) stage_load(ci + 1);
      const char* buf = smem + L::stage + (ci & 1) * (2 * L::plane_bytes);
#pragma unroll 1
      for (int ks = 0; ks < kSplChunkPts / 4; ++ks) {
        const int plc = 4 * ks + q;                              // point of the chunk
        const int pl = ci * kSplChunkPts + plc;                  // point of the run
        if (uni(ci * kSplChunkPts + 4 * ks) >= count) break;     // (runs of 1, 2, 4 points: uniform)
        const bool live = pl < count;
        const int pt = first + (live ? pl : 0);
        // A: slots 0..3 and 4..7 of this lane's point; slot 7 and points beyond the run read the zero row
        const char* a_lo4 = live ? buf + (plc * LIST_N_STENCIL + tr_r) * RB : smem + L::zero;
        const char* a_hi4 = (live && tr_r < 3) ? buf + (plc * LIST_N_STENCIL + 4 + tr_r) * RB : smem + L::zero;
        const int lo_plane = live ? L::plane_bytes : 0;                       // (the zero row has no second plane)
        const int lo_plane_hi4 = (live && tr_r < 3) ? L::plane_bytes : 0;
        // B: the 7 weights of (point, slot) at this lane's voxel, per owned tile, split hi + lo in bf16
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
        bf16x8 bhi[2], blo[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          if (i == 0 ? !own0 : !own1) continue;
          const float cx = live ? hx[i][0] : 0.f;                  // (every slot's weight has a centre factor)
          const float yz = hy[i][0] * hz[i][0], xz = cx * hz[i][0], xy = cx * hy[i][0];
          const float w[8] = {cx * yz, hx[i][1] * yz, hx[i][2] * yz, hy[i][1] * xz, hy[i][2] * xz, hz[i][1] * xy, hz[i][2] * xy, 0.f};
          unsigned hi[4], lo[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            hi[e] = pk_bf2(w[2 * e], w[2 * e + 1]);
            lo[e] = pk_bf2(w[2 * e] - bf2f((unsigned short)(hi[e] & 0xffffu)), w[2 * e + 1] - bf2f((unsigned short)(hi[e] >> 16)));
          }
          bhi[i] = __builtin_bit_cast(bf16x8, make_uint4(hi[0], hi[1], hi[2], hi[3]));
          blo[i] = __builtin_bit_cast(bf16x8, make_uint4(lo[0], lo[1], lo[2], lo[3]));
        }
        // the channel tiles in two halves (their hi and lo fragments are 32 registers per half)
#pragma unroll
        for (int th = 0; th < 2; ++th) {
          bf16x8 ah[NT / 2], al[NT / 2];
#pragma unroll
          for (int tt = 0; tt < NT / 2; ++tt) {
            const int t = th * (NT / 2) + tt;
            const s16x4 h0 = tr_read16(a_lo4 + aoff[t]), h1 = tr_read16(a_hi4 + aoff[t]);
            const s16x4 l0 = tr_read16(a_lo4 + lo_plane + aoff[t]), l1 = tr_read16(a_hi4 + lo_plane_hi4 + aoff[t]);
            ah[tt] = __builtin_bit_cast(bf16x8, (s16x8){h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]});
            al[tt] = __builtin_bit_cast(bf16x8, (s16x8){l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]});
          }
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            if (i == 0 ? !own0 : !own1) continue;
#pragma unroll
            for (int tt = 0; tt < NT / 2; ++tt) {
              const int t = th * (NT / 2) + tt;
              acc[i][t] = mfma16<0>(al[tt], bhi[i], acc[i][t]);
              acc[i][t] = mfma16<0>(ah[tt], blo[i], acc[i][t]);
              acc[i][t] = mfma16<0>(ah[tt], bhi[i], acc[i][t]);
            }
          }
        }
      }
    }
    __syncthreads();                   // every wave is done with the staging images: the fp32 box takes their place
    // D: column = box row (lane & 15), rows 4 q + reg of tile t = channels 32 (t >> 1) + 8 q + 4 (t & 1) + reg -> one 16-B
    // piece of 4 channels per tile: chunk 8 (t >> 1) + 2 q + (t & 1) of the row's 32, XOR (row & 31)
    float* box = (float*)(smem + L::stage);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (i == 0 ? !own0 : !own1) continue;
      const int v = 16 * (wave + 4 * i) + col;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int chunk = (8 * (t >> 1) + 2 * q + (t & 1)) ^ (v & 31);
        *(float4*)(box + v * kSplC + chunk * 4) = make_float4(acc[i][t][0], acc[i][t][1], acc[i][t][2], acc[i][t][3]);
      }
    }
    __syncthreads();
    // flush: one box row per wave and pass, two instructions of 64 channels (256 contiguous bytes) each
    {
      float* base = (float*)gv.data + (int64_t)rb_b * gv.image_stride;
#pragma unroll 1
      for (int v = wave; v < rows; v += 4) {
        const int yz = (v * inv_nx) >> 16, ix = v - yz * nx;
        const int iz = (yz * inv_ny) >> 16, iy = yz - iz * ny;
        float* dst = base + ((int64_t)((loz + iz) * H + (loy + iy)) * W + (lox + ix)) * kSplC;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int ch = 64 * h + lane;
          const float val = box[v * kSplC + (((ch >> 2) ^ (v & 31)) << 2) + (ch & 3)];
          if (val != 0.f) atomicAdd(dst + ch, val * inv_s);
        }
      }
    }
    first += count;
    if (first < kSplPts) __syncthreads();                       // the next run stages over the box
  }
}

// a window level (stencil shorter than a voxel), fp32 dX, 128 channels
bool scatter_box_split_eligible(const ScatterParams& sp, const ListVoxLevel& gv, int col_off) {
  static const bool off = [] { const char* e = getenv("LIST_SCATTER_BOX"); return e && e[0] == '0' && e[1] == 0; }();
  if (off || sp.dx_f16 || gv.C != kSplC) return false;
  if ((col_off % 4) != 0 || (sp.g.Kp % 4) != 0) return false;
  if (gv.W > 255 || gv.H > 255 || gv.D > 255) return false;   // 8-bit coordinates in the run records
  return (sp.g.rows % kSplPts) == 0;
}

hipError_t launch_scatter_vox_box_split(const ScatterParams& sp, const ListVoxLevel& gv, int col_off, hipStream_t s) {
  hipLaunchKernelGGL(k_scatter_vox_box_split, dim3((unsigned)(sp.g.rows / kSplPts)), dim3(256), 0, s, sp, gv, col_off);
  return hipGetLastError();
}

}  // namespace list
