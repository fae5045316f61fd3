// Coarse voxel levels (stencil shorter than a voxel: the 16^3 and 8^3 x 128-channel levels, 58 % of all tap
// contributions) on the matrix cores: network/modules.py:256-265 for those levels as small GEMMs.
//
// A stencil sample (point p, stencil point j) is a weighted sum of the 2 x 2 x 2 voxels around it,
//   out[(p,j)][c] = sum_v Wt[(p,j)][v] * V[v][c],   Wt = hx[vx] * hy[vy] * hz[vz]   (two non-zeros per axis).
// Samples that share their "window" -- the 2 x 2 cells (y0, y0+1) x (z0, z0+1) and the 8 x-slots starting at
// 4 * (x0 >> 2) -- share the 32 voxel rows V[32][C] of that window, so 16 of them are one MFMA column tile:
//   D^T[c][s] = sum_k V^T[c][k] * Wt^T[k][s],  k = 8 * (2 * dz + dy) + x-slot,  v_mfma_f32_16x16x32_f16,
// with the weights formed per lane directly in the B-fragment layout (8 consecutive k = the 8 x-slots of one
// (dy, dz) cell: the product (hy * hz) times the two non-zero x weights, split hi + lo in fp16 so that the
// products are exact in the fp32 accumulator: same interpolation arithmetic as the scalar kernels up to the
// order of the fp32 sums) and V^T read with ds_read_b64_tr_b16 from a channels-last LDS copy of the voxel box
// that a run of Morton-consecutive points touches (ONE coalesced copy instead of 32 tap requests per point).
//
// Every sample's arithmetic depends on nothing but its own window and weights -- the k position of every voxel
// is a function of absolute voxel coordinates, zero-weight slots add exact zeros -- so results do not depend on
// which points share a workgroup, a run or a tile: sharded == unsharded bit for bit still holds.
// Non-finite voxels inside a window turn 0 * inf into NaN for samples that do not touch them; such rows are
// flagged by fc_0's probe and redone by k_gather_fixup with the reference's skip semantics, like before.
//
// Workgroup = 64 consecutive rows (Morton order), 256 threads:
//   1. per point: the three per-axis weight records of the centre / -d / +d coordinate (AxisW), its tap range;
//   2. wave 0 cuts the 64 points into aligned power-of-two runs whose voxel box fits the LDS box (a segment
//      tree over the tap ranges by wave shuffles; a single point always fits: 4 x 4 x 4);
//   3. per run: the box is copied to LDS (XOR-swizzled 16-B chunks: the transposed reads are bank-conflict
//      free), the run's samples are counting-sorted by window key in LDS and cut into tiles of <= 16 samples,
//      and the four waves walk the tiles: weights -> 16 MFMAs (8 channel tiles x hi / lo) -> 16-B stores of
//      8 consecutive channels per lane straight into the sample's X row.
#include "list_common.h"
#include "point_math.h"
#include "box_partition.h"

namespace list {

constexpr int kBoxPts = 64;                                  // points per workgroup
constexpr int kBoxSamples = kBoxPts * LIST_N_STENCIL;        // 448
constexpr int kBoxMaxKeys = 512;                             // window keys a run may have
#ifndef LIST_BOX_NB
#define LIST_BOX_NB 2                                        // tiles per turn of a wave
#endif
#ifndef LIST_BOX_ROWS_BIG
#ifdef LIST_BOX_XPOSE
#define LIST_BOX_ROWS_BIG 224                                // LDS box rows, levels wider than 8 voxels (2 workgroups per CU)
#else
#define LIST_BOX_ROWS_BIG 256
#endif
#endif
#ifndef LIST_BOX_ROWS_SMALL
#ifdef LIST_BOX_XPOSE
#define LIST_BOX_ROWS_SMALL 120                              // (3 workgroups per CU)
#else
#define LIST_BOX_ROWS_SMALL 128
#endif
#endif

// diagnostic build only (-DLIST_BOX_STAMPS): cycles per phase, summed over the workgroups by wave 0 / lane 0 into a
// buffer of their own (list_debug_box_stamps); no stamp executes in the shipped kernel
#ifdef LIST_BOX_STAMPS
__device__ unsigned long long g_box_stamps[2][4096][16];      // [small / big box][workgroup][slot]: own words, no atomics
#define BOX_STAMP(slot)                                                                                     \
  do {                                                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    unsigned long long t_;                                                                                  \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                             \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    if (tid == 0) st_[slot] += t_ - t_prev_;                                                                \
    t_prev_ = t_;                                                                                           \
  } while (0)
#define BOX_STAMP_INIT()                                                                                    \
  unsigned long long t_prev_, st_[16] = {0};                                                                \
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_prev_)::"memory")
#else
#define BOX_STAMP(slot) do {} while (0)
#define BOX_STAMP_INIT() do {} while (0)
#endif

template <int C, int MAXROWS> struct BoxLds {
  static constexpr int kRowBytes = 2 * C;
  static constexpr int box = 0;
  static constexpr int ptab = box + MAXROWS * kRowBytes;                 // AxisW [64][3 axes][3 variants]
  static constexpr int run = ptab + kBoxPts * 9 * (int)sizeof(AxisW);    // RunBox [64], indexed by first point
  static constexpr int keyinfo = run + kBoxPts * (int)sizeof(RunBox);    // int [512]: counters, then sample offsets
  static constexpr int tileinfo = keyinfo + kBoxMaxKeys * 4;             // unsigned [448]
  static constexpr int sorted = tileinfo + kBoxSamples * 4;              // short [448]: sample ids in key order
  static constexpr int pbox = sorted + kBoxSamples * 2;                  // int [64][4]: tap range per axis, image
  static constexpr int misc = pbox + kBoxPts * 16;                       // int [8]: tile count
#ifdef LIST_BOX_XPOSE
  static constexpr int xpose = misc + 64;                                // 4 waves x [8 samples][256 B]: store staging
  static constexpr int total = xpose + 4 * 8 * kRowBytes;
#else
  static constexpr int total = misc + 64;
#endif
};

// grid = rows / 64, block = 256
template <int C, int MAXROWS>
__global__ __launch_bounds__(256) void k_gather_vox_box(GatherParams g, ListVoxLevel lv, int col_off) {
  using L = BoxLds<C, MAXROWS>;
  static_assert(C % 128 == 0 && C <= 256, "channel tiles of 32 and a 16-chunk swizzle");
  static_assert(MAXROWS >= 64 && MAXROWS <= 256, "a single point's 4 x 4 x 4 box must fit; row indices are 8-bit");
  constexpr int NT = C / 16;                                  // 16-channel MFMA row tiles
  constexpr int RB = L::kRowBytes;
  constexpr int CH = RB / 16;                                 // 16-B chunks per voxel row
  constexpr int RPP = 256 / CH;                               // box rows per pass of the workgroup
  constexpr int NL = (MAXROWS + RPP - 1) / RPP;               // 16-B loads per thread for a full box
  __shared__ __attribute__((aligned(16))) char smem[L::total];
  AxisW* ptab = (AxisW*)(smem + L::ptab);
  RunBox* runs = (RunBox*)(smem + L::run);
  int* keyinfo = (int*)(smem + L::keyinfo);
  unsigned* tileinfo = (unsigned*)(smem + L::tileinfo);
  short* sorted = (short*)(smem + L::sorted);
  int* pbox = (int*)(smem + L::pbox);
  int* misc = (int*)(smem + L::misc);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = uni(tid >> 6);
  const int blk = xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int64_t row0 = (int64_t)blk * kBoxPts;
  const int W = lv.W, H = lv.H, D = lv.D;

  BOX_STAMP_INIT();
  // ---- 1a. waves 0..2: axis `wave` of the 64 points -- weight records of the centre / -d / +d coordinate -----------
  if (wave < 3) {
    const Pt p = load_point(g, (int)row0 + lane);
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
    for (int k = lane; k < kBoxMaxKeys; k += 64) keyinfo[k] = 0;
  }
  BOX_STAMP(0);
  __syncthreads();
  // ---- 1b. wave 0: aligned power-of-two runs whose box fits (segment tree over the tap ranges) ----------------------
  if (wave == 0) {
    const int4 pb = *(const int4*)(pbox + lane * 4);
    const bool valid = pb.w >= 0;
    SegBox sb;
    sb.f0 = valid ? (unsigned)((pb.x & 255) | ((pb.y & 255) << 16)) : 0x7fff7fffu;
    sb.f1 = valid ? (unsigned)((pb.z & 255) | ((255 - (pb.x >> 8)) << 16)) : 0x7fff7fffu;
    sb.f2 = valid ? (unsigned)((255 - (pb.y >> 8)) | ((255 - (pb.z >> 8)) << 16)) : 0x7fff7fffu;
    sb.bmin = valid ? pb.w : INT_MAX;
    sb.nbmax = valid ? ~pb.w : INT_MAX;
    // a segment that fits contains only segments that fit: the largest fitting level is the same for all its lanes
    int level = 0;
    SegBox best = sb;
#define LIST_SEG_STAGE(S)                                              \
    seg_merge<S>(sb);                                                  \
    if (level == S && seg_fits(sb, MAXROWS, kBoxMaxKeys)) { level = S + 1; best = sb; }
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
  BOX_STAMP(1);
  __syncthreads();
  BOX_STAMP(2);

  const unsigned short* __restrict__ vsrc = (const unsigned short*)lv.data;
  unsigned short* __restrict__ xh = g.x_hi;
  const int q = lane >> 4, col = lane & 15;                  // MFMA lane roles: k block / sample column
  const int tr_r = (lane >> 2) & 3, tr_p = lane & 3;         // transposed read: row within the 4-row block, 4-channel group

  int first = 0;
#pragma unroll 1
  while (first < kBoxPts) {
    const RunBox rb = runs[first];
    const int count = uni(rb.count), rb_b = uni(rb.b), rlo = uni(rb.lo), rn = uni(rb.n);
    const int lox = rlo & 255, loy = (rlo >> 8) & 255, loz = rlo >> 16;
    const int nx = rn & 255, ny = (rn >> 8) & 255, nz = rn >> 16;
    const int rows = nx * ny * nz;
    const int fw0 = lox >> 2;
    const int nfw = rows ? ((lox + nx - 1) >> 2) - fw0 + 1 : 0;
    const int nkeys = nfw * ny * nz;
    const int inv_nx = (65536 + nx - 1) / (nx > 0 ? nx : 1), inv_ny = (65536 + ny - 1) / (ny > 0 ? ny : 1);
    const int inv_nz = (65536 + nz - 1) / (nz > 0 ? nz : 1);

    // ---- 2a. the voxel box: requested now (16-B chunks, lanes over channels: coalesced 256-B rows), landed in LDS
    //          behind the bucketing below ---------------------------------------------------------------------------
    uint4 bv[NL];
    int bdst[NL];
    {
      const int chunk = tid & (CH - 1);
      const int64_t ibase = (int64_t)rb_b * lv.image_stride;
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        const int r = i * RPP + tid / CH;
        bdst[i] = -1;
        if (r < rows) {
          const int yz = (r * inv_nx) >> 16, ix = r - yz * nx;
          const int iz = (yz * inv_ny) >> 16, iy = yz - iz * ny;
          const int64_t src = ibase + ((int64_t)((loz + iz) * H + (loy + iy)) * W + (lox + ix)) * C + chunk * 8;
          bv[i] = *(const uint4*)(vsrc + src);
          // physical position: chunk ^ ((ix & 3) << 2), 8-B halfs swapped in odd (iy + iz) cells: the 32 lanes of a
          // transposed read (4 x-slots x 2 cells x 4 channel groups) then hit 32 distinct 8-B bank slots
          const int par = (iy + iz) & 1;
          bdst[i] = (r * RB + ((chunk ^ ((ix & 3) << 2)) << 4)) | (par << 30);
        }
      }
    }

    // ---- 2b. window key of every sample (counters are zero); padding rows: their X columns are zero ----------------
    const int nsamp = count * LIST_N_STENCIL;
    int my_key[2], my_slot[2], my_sid[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int s = tid + 256 * i;
      my_key[i] = -1; my_slot[i] = 0; my_sid[i] = 0;
      if (s >= nsamp) continue;
      const int pl = s / LIST_N_STENCIL, j = s - pl * LIST_N_STENCIL, pt = first + pl;
      if ((int)row0 + pt >= g.n_valid) {                      // padding row
        uint4* o = (uint4*)(xh + (row0 + pt) * g.Kp + col_off + j * C);
#pragma unroll 1
        for (int c8 = 0; c8 < C / 8; ++c8) o[c8] = make_uint4(0u, 0u, 0u, 0u);
        continue;
      }
      const int x0 = ptab[(pt * 3 + 0) * 3 + variant_of(0, j)].i0;
      const int y0 = ptab[(pt * 3 + 1) * 3 + variant_of(1, j)].i0;
      const int z0 = ptab[(pt * 3 + 2) * 3 + variant_of(2, j)].i0;
      const int key = (((x0 >> 2) - fw0) * nz + (z0 - loz)) * ny + (y0 - loy);
      my_key[i] = key;
      my_sid[i] = pt * 8 + j;
      my_slot[i] = atomicAdd(&keyinfo[key], 1);
    }
    __syncthreads();
    BOX_STAMP(3);

    // ---- 2c. wave 0: exclusive scan over the keys -> sample offsets, tiles; meanwhile the box lands -----------------
    if (wave == 0) {
      int carry = 0;                                          // tiles << 16 | samples
#pragma unroll 1
      for (int k0 = 0; k0 < nkeys; k0 += 64) {
        const int key = k0 + lane;
        const int cnt = key < nkeys ? keyinfo[key] : 0;
        const int ntl = (cnt + 15) >> 4;
        const int mine = (ntl << 16) | cnt;
        int incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
          const int v = __shfl_up(incl, off);
          if (lane >= off) incl += v;
        }
        const int at = carry + incl - mine;
        carry += __shfl(incl, 63);
        if (key < nkeys) {
          keyinfo[key] = at & 0xffff;
          if (cnt) {
            // tiles of this key: start | n << 9 | y0 << 14 | z0 << 19 | window << 24
            const int yz = (key * inv_ny) >> 16, y0 = key - yz * ny;
            const int fw = (yz * inv_nz) >> 16, z0 = yz - fw * nz;
            int tile = at >> 16, start = at & 0xffff;
            for (int i = 0; i < ntl; ++i, ++tile, start += 16) {
              const int n = cnt - 16 * i < 16 ? cnt - 16 * i : 16;
              tileinfo[tile] = (unsigned)start | ((unsigned)n << 9) | ((unsigned)y0 << 14) | ((unsigned)z0 << 19) |
                               ((unsigned)fw << 24);
            }
          }
        }
      }
      if (lane == 0) misc[0] = carry >> 16;                   // number of tiles
    }
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      if (bdst[i] < 0) continue;
      const uint4 w = (bdst[i] >> 30) & 1 ? make_uint4(bv[i].z, bv[i].w, bv[i].x, bv[i].y) : bv[i];
      *(uint4*)(smem + L::box + (bdst[i] & 0x3fffffff)) = w;
    }
    __syncthreads();
    BOX_STAMP(4);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      if (my_key[i] >= 0) sorted[keyinfo[my_key[i]] + my_slot[i]] = (short)my_sid[i];
    __syncthreads();
    BOX_STAMP(5);
    for (int k = tid; k < nkeys; k += 256) keyinfo[k] = 0;    // counters of the next run (nothing below reads them)

    // ---- 2d. tiles: <= 16 samples of one window; two tiles per turn (independent chains of LDS look-ups) -----------
    const int ntiles = uni(misc[0]);
    constexpr int NB = LIST_BOX_NB;
#ifdef LIST_BOX_NO_TILES
    if (g.Kp < 0)
#endif
#pragma unroll 1
    for (int T = wave; T < ntiles; T += 4 * NB) {
      unsigned bh[NB][4], bl[NB][4];
      int sid[NB], abase[NB][2], aswz[NB][2];
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int Tb = T + 4 * b < ntiles ? T + 4 * b : T;    // (a turn's second tile may not exist: recomputed, not stored)
        const unsigned ti = (unsigned)uni((int)tileinfo[Tb]);
        const int start = ti & 511, n = (ti >> 9) & 31, y0 = (ti >> 14) & 31, z0 = (ti >> 19) & 31, fw = ti >> 24;
        // weights of this lane's sample for the 8 x-slots of cell q = (dy, dz): two non-zeros, hi + lo in fp16
        // (no branch: the look-up chains of the turn's tiles overlap; a column beyond n repeats the last sample
        // with weight 0)
        {
          const bool live = col < n;
          sid[b] = sorted[start + (live ? col : n - 1)];
          const int pt = sid[b] >> 3, j = sid[b] & 7;
          const AxisW fx = ptab[(pt * 3 + 0) * 3 + variant_of(0, j)];
          const AxisW fy = ptab[(pt * 3 + 1) * 3 + variant_of(1, j)];
          const AxisW fz = ptab[(pt * 3 + 2) * 3 + variant_of(2, j)];
          const float s = live ? ((q & 1) ? fy.w1 : fy.w0) * ((q >> 1) ? fz.w1 : fz.w0) : 0.f;
          const float wa = s * fx.w0, wb = s * fx.w1;
          const unsigned hi2 = pk_h2(wa, wb);
          const float ra = wa - h2f((unsigned short)(hi2 & 0xffffu)), rb2 = wb - h2f((unsigned short)(hi2 >> 16));
          const unsigned lo2 = pk_h2(ra, rb2);
          const int phi = fx.i0 & 3;                          // x-slot of the base tap in its window
          // slots phi, phi + 1 of the 8-slot fragment (4 registers of two halfs)
          bh[b][0] = phi == 0 ? hi2 : (phi == 1 ? hi2 << 16 : 0u);
          bh[b][1] = phi == 2 ? hi2 : (phi == 3 ? hi2 << 16 : (phi == 1 ? hi2 >> 16 : 0u));
          bh[b][2] = phi == 3 ? hi2 >> 16 : 0u;
          bh[b][3] = 0u;
          bl[b][0] = phi == 0 ? lo2 : (phi == 1 ? lo2 << 16 : 0u);
          bl[b][1] = phi == 2 ? lo2 : (phi == 3 ? lo2 << 16 : (phi == 1 ? lo2 >> 16 : 0u));
          bl[b][2] = phi == 3 ? lo2 >> 16 : 0u;
          bl[b][3] = 0u;
          if (!live) sid[b] = -1;
        }
        if (T + 4 * b >= ntiles) sid[b] = -1;
        // V^T fragments: this lane addresses row tr_r of the 4-row block, cell q, channel group tr_p.  Cells and slots
        // beyond the box carry weight 0 for every sample of the run: they re-read the box's border rows
        const int iy = min(y0 + (q & 1), ny - 1), iz = min(z0 + (q >> 1), nz - 1);
        const int par = (iy + iz) & 1;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int ix = min(max(4 * (fw0 + fw) + 4 * h + tr_r - lox, 0), nx - 1);
          abase[b][h] = L::box + ((iz * ny + iy) * nx + ix) * RB + (tr_p << 4);
          aswz[b][h] = ((ix & 3) << 6) | (par << 3);
        }
      }
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        if (b > 0 && T + 4 * b >= ntiles) break;
        const f16x8 bhi = __builtin_bit_cast(f16x8, make_uint4(bh[b][0], bh[b][1], bh[b][2], bh[b][3]));
        const f16x8 blo = __builtin_bit_cast(f16x8, make_uint4(bl[b][0], bl[b][1], bl[b][2], bl[b][3]));
        // channels of row tile t: 32 (t >> 1) + 8 p + 4 (t & 1) + 0..3 for 4-channel group p -- so that a lane's
        // accumulators of tiles 2u, 2u + 1 are 8 consecutive channels.  All 16 transposed reads first, then the MFMAs
        s16x4 a0[NT], a1[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int ct = ((t >> 1) << 6) | ((t & 1) << 3);
          a0[t] = tr_read16(smem + abase[b][0] + (ct ^ aswz[b][0]));
          a1[t] = tr_read16(smem + abase[b][1] + (ct ^ aswz[b][1]));
        }
        f32x4v acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const f16x8 a = __builtin_bit_cast(f16x8, (s16x8){a0[t][0], a0[t][1], a0[t][2], a0[t][3], a1[t][0], a1[t][1], a1[t][2], a1[t][3]});
          f32x4v c = {0.f, 0.f, 0.f, 0.f};
          c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bhi, c, 0, 0, 0);
          acc[t] = c;
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const f16x8 a = __builtin_bit_cast(f16x8, (s16x8){a0[t][0], a0[t][1], a0[t][2], a0[t][3], a1[t][0], a1[t][1], a1[t][2], a1[t][3]});
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, blo, acc[t], 0, 0, 0);
        }
        // D^T: column = sample (lane & 15), rows 4 q + reg of tile t = channels 32 (t >> 1) + 8 q + 4 (t & 1) + reg.
        // Plain stores: the four lanes of a sample cover 64 B per instruction and the partial lines merge in L2
        // (non-temporal stores of such pieces took the kernel from 0.085 to 0.30 ms)
#ifdef LIST_BOX_XPOSE
        {
          // whole rows per store instruction: the tile turns through a wave-private LDS image, 8 samples at a time
          // ([sample][16-B chunk ^ sample]: conflict-free both ways; DS operations of one wave execute in order), then
          // every 16 lanes hold one sample's 256 B -> non-temporal stores of full rows like the scalar kernels'
          typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
          char* tw = smem + L::xpose + wave * (8 * RB);
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            if ((col >> 3) == hh) {
#pragma unroll
              for (int u = 0; u < NT / 2; ++u) {
                const uint2 lo = half4_inrange(make_float4(acc[2 * u][0], acc[2 * u][1], acc[2 * u][2], acc[2 * u][3]));
                const uint2 hi = half4_inrange(make_float4(acc[2 * u + 1][0], acc[2 * u + 1][1], acc[2 * u + 1][2], acc[2 * u + 1][3]));
                *(uint4*)(tw + (col & 7) * RB + (((4 * u + q) ^ (col & 7)) << 4)) = make_uint4(lo.x, lo.y, hi.x, hi.y);
              }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // lanes read what other lanes of the wave stored
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
              const int r8 = 4 * k2 + (lane >> 4), chunk = lane & 15;
              const uint4 v = *(const uint4*)(tw + r8 * RB + ((chunk ^ r8) << 4));
              const int sd = __shfl(sid[b], 8 * hh + r8);
              if (sd >= 0) {
                unsigned short* dst = xh + (row0 + (sd >> 3)) * g.Kp + col_off + (sd & 7) * C + 8 * chunk;
                __builtin_nontemporal_store((u32x4){v.x, v.y, v.z, v.w}, (u32x4*)dst);
              }
            }
          }
        }
#else
        if (sid[b] >= 0) {
          const int pt = sid[b] >> 3, j = sid[b] & 7;
          unsigned short* dst = xh + (row0 + pt) * g.Kp + col_off + j * C + 8 * q;
          typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#pragma unroll
          for (int u = 0; u < NT / 2; ++u) {
            const uint2 lo = half4_inrange(make_float4(acc[2 * u][0], acc[2 * u][1], acc[2 * u][2], acc[2 * u][3]));
            const uint2 hi = half4_inrange(make_float4(acc[2 * u + 1][0], acc[2 * u + 1][1], acc[2 * u + 1][2], acc[2 * u + 1][3]));
#if defined(LIST_BOX_NO_STORE)
            asm volatile("" :: "v"(lo.x), "v"(lo.y), "v"(hi.x), "v"(hi.y));
#elif defined(LIST_BOX_NT_STORE)
            __builtin_nontemporal_store((u32x4){lo.x, lo.y, hi.x, hi.y}, (u32x4*)(dst + 32 * u));
#elif defined(LIST_BOX_SC1_STORE)
            {
              const u32x4 val = {lo.x, lo.y, hi.x, hi.y};
              asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst + 32 * u), "v"(val) : "memory");
            }
#else
            *(u32x4*)(dst + 32 * u) = (u32x4){lo.x, lo.y, hi.x, hi.y};
#endif
          }
        }
#endif
      }
    }
    first += count;
    BOX_STAMP(6);
    if (first < kBoxPts) __syncthreads();                     // the next run overwrites the box and the lists
    BOX_STAMP(7);
#ifdef LIST_BOX_STAMPS
    if (tid == 0) { st_[8] += 1; st_[9] += ntiles; st_[10] += rows; st_[11] += nkeys; }
#endif
  }
#ifdef LIST_BOX_STAMPS
  if (tid == 0 && blockIdx.x < 4096) {
    st_[12] = 1;
    for (int i = 0; i < 16; ++i) g_box_stamps[MAXROWS == LIST_BOX_ROWS_BIG ? 1 : 0][blockIdx.x][i] = st_[i];
  }
#endif
}

// near level with fp16 maps and an fp16 feature matrix; false: not taken (the caller falls back to k_gather_vox_near)
bool gather_box_eligible(const GatherParams& g, const ListVoxLevel& lv, int col_off) {
  static const bool off = [] { const char* e = getenv("LIST_GATHER_BOX"); return e && e[0] == '0' && e[1] == 0; }();
  if (off) return false;
  if (g.fmt != FMT_FP16 || lv.dtype != LIST_MAP_F16 || lv.C != 128) return false;
  if ((col_off % 8) != 0 || (g.Kp % 8) != 0 || (lv.image_stride % 8) != 0) return false;
  if (lv.W > 31 || lv.H > 31 || lv.D > 31) return false;       // 5-bit cell indices in the tile records
  return (g.rows % kBoxPts) == 0;
}

hipError_t launch_gather_vox_box(const GatherParams& g, const ListVoxLevel& lv, int col_off, hipStream_t s, int order) {
  // the 8^3 level's boxes are small: a 128-row box leaves room for three workgroups per CU
  const int big = lv.W > lv.H ? (lv.W > lv.D ? lv.W : lv.D) : (lv.H > lv.D ? lv.H : lv.D);
  if (big > 8)
    LIST_LAUNCH((k_gather_vox_box<128, LIST_BOX_ROWS_BIG>), dim3(g.rows / kBoxPts), dim3(256), 0, s, order, g, lv, col_off);
  else
    LIST_LAUNCH((k_gather_vox_box<128, LIST_BOX_ROWS_SMALL>), dim3(g.rows / kBoxPts), dim3(256), 0, s, order, g, lv, col_off);
  return hipGetLastError();
}

#ifdef LIST_BOX_STAMPS
extern "C" int list_debug_box_stamps(unsigned long long* out, int reset) {     // out: [2][4096][16]
  (void)reset;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_box_stamps), sizeof(g_box_stamps)) != hipSuccess) return -1;
  return 0;
}
#endif

}  // namespace list
