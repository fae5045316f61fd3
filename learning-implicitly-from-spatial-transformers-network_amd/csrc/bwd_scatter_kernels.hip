// Backward of the gathers: dX [rows][Kp] (gradient of the feature matrix, gather order) ->
//   * gradients of the channels-last voxel levels     (adjoint of F.grid_sample 3-D, modules.py:263-265)
//   * gradient of the prepared 137^2 perceptual map   (adjoint of F.grid_sample 2-D, modules.py:46-52)
//   * gradient of trans_mat                           (through modules.py:37-45: matmul, divide, clamp)
//   * gradients of the five encoder maps              (adjoint of F.interpolate, modules.py:26-35)
// HBM / atomic-rate bound: global float atomics run at ~1.3 TB/s chip-wide on MI355X whatever the kernel does,
// so the design goal is FEWER atomic bytes, shaped as runs along the channel axis.  Per voxel level one of
//   * global atomics, lanes over channels            (fine levels: 56 distinct taps per point, nothing to merge)
//   * voxel-side gather over cell-sorted samples      (dense middle level: written once, no atomics)
//   * LDS windows over runs of Morton-ordered points  (coarse levels: summed on chip without atomics -- every
//                                                     thread owns one channel column --, flushed once)
// and for the perceptual map no atomics at all: the points are already in pixel order (forward's second
// sort), so every map pixel GATHERS the contributions of the <= 4 pixel cells around it and is written once.
#include <limits.h>
#include <string.h>

#include "list_common.h"
#include "point_math.h"

namespace list {

template <int DXH>
__device__ __forceinline__ float dx_at(const void* __restrict__ dx, int64_t i) {
  return DXH ? h2f(((const unsigned short*)dx)[i]) : ((const float*)dx)[i];
}

__device__ __forceinline__ void stencil_rt(const Pt& p, int j, float& x, float& y, float& z) {
  x = p.x + (j == 1 ? -kDisp : j == 2 ? kDisp : 0.f);
  y = p.y + (j == 3 ? -kDisp : j == 4 ? kDisp : 0.f);
  z = p.z + (j == 5 ? -kDisp : j == 6 ? kDisp : 0.f);
}

// ---- voxel levels ----------------------------------------------------------------------------------------
constexpr int kScatterRows = 64;           // points per workgroup (consecutive rows = one Morton neighbourhood)

// Straight to memory: C lanes per (point, stencil point) item, one 4*C-byte atomic run per tap.  This is
// what the levels whose stencil spreads over many voxels use (128^3 .. 32^3): their 56 taps per point are
// all distinct and neighbouring points share few of them, so the cost is the chip's atomic byte rate.
template <int C, int DXH, int T = 256>
__device__ __forceinline__ void scatter_direct(const ScatterParams& sp, const ListVoxLevel& gv, int col_off,
                                               const Pt* __restrict__ pts, int r_begin, int r_end,
                                               int64_t row0, float inv_s) {
  constexpr int per_pass = T / C;
  const int tid = threadIdx.x;
  const int c = tid % C;
  float* __restrict__ gout = (float*)gv.data;
  for (int it = r_begin * LIST_N_STENCIL + tid / C; it < r_end * LIST_N_STENCIL; it += per_pass) {
    const int r = it / LIST_N_STENCIL, j = it - r * LIST_N_STENCIL;
    const Pt p = pts[r];
    if (!p.valid) continue;
    float x, y, z;
    stencil_rt(p, j, x, y, z);
    const Taps t = make_taps(x, y, z, C, gv.D, gv.H, gv.W);
    const float gval = dx_at<DXH>(sp.dx, (row0 + r) * sp.g.Kp + col_off + j * C + c) * inv_s;
    float* base = gout + (int64_t)p.b * gv.image_stride + c;
#pragma unroll
    for (int k = 0; k < 8; ++k) atomicAdd(base + t.o[k], t.w[k] * gval);
  }
}

// Persistent grid (workgroups walk the 64-row blocks).  Alone, two workgroups per CU reach the atomic rate
// (512: 1.73 ms for the three fine levels; 128: +28 %, 64: 2.1x -- the rate is partly a per-CU issue
// limit).  Beside other kernels (ListQueryGradArgs.aux_streams) a small grid is better for the whole: the
// atomics of a CU fill its vector-memory queue, and every load of a neighbouring gather kernel waits behind
// them (backward 5.92 ms with 512 workgroups, 5.57 ms with 128).
#ifndef LIST_DIRECT_GRID_FORKED
#define LIST_DIRECT_GRID_FORKED 128      // (re-measured in round 3 with dW0 capped at 194 registers: DESIGN 5b)
#endif
constexpr int kDirectGrid = 512, kDirectGridForked = LIST_DIRECT_GRID_FORKED;

template <int C, int DXH>
__global__ __launch_bounds__(256) void k_scatter_vox(ScatterParams sp, ListVoxLevel gv, int col_off, int nblocks) {
  __shared__ Pt pts[kScatterRows];
  const float inv_s = sp.scale[1];
  for (int blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
    __syncthreads();
    if (threadIdx.x < kScatterRows) pts[threadIdx.x] = load_point(sp.g, blk * kScatterRows + threadIdx.x);
    __syncthreads();
    scatter_direct<C, DXH>(sp, gv, col_off, pts, 0, kScatterRows, (int64_t)blk * kScatterRows, inv_s);
  }
}

// fp16 operands, levels whose tap run is two or more 64-B atomic requests in fp32 (C >= 32): the direct form's
// time follows the atomic BYTES there (same 9 M tap runs per launch: C = 16 / 64 B 0.51 ms, C = 32 / 128 B 0.89 ms),
// so the runs are added as packed halfs (global_atomic_pk_add_f16, lanes over channel PAIRS) into a zeroed fp16 image
// of the level kept at the gradient scale s (the dX operand is s * dX already, fp16), and one streaming pass
// writes (1/s) * image as the fp32 gradient (which then needs no memset).  A voxel of these sparse levels sums a few
// contributions, each rounded to 11 bits -- far inside the fp16 mode's gradient noise; the fp32-grade
// mode never takes this form.
typedef _Float16 half2v __attribute__((ext_vector_type(2)));

template <int C>
__global__ __launch_bounds__(256) void k_scatter_vox_h2(ScatterParams sp, ListVoxLevel gv, int col_off, int nblocks,
                                                        _Float16* __restrict__ img16) {
  __shared__ Pt pts[kScatterRows];
  constexpr int CP = C / 2, per_pass = 256 / CP;
  const int tid = threadIdx.x, cp = tid % CP;
  for (int blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
    __syncthreads();
    if (threadIdx.x < kScatterRows) pts[threadIdx.x] = load_point(sp.g, blk * kScatterRows + threadIdx.x);
    __syncthreads();
    const int64_t row0 = (int64_t)blk * kScatterRows;
    for (int it = tid / CP; it < kScatterRows * LIST_N_STENCIL; it += per_pass) {
      const int r = it / LIST_N_STENCIL, j = it - r * LIST_N_STENCIL;
      const Pt p = pts[r];
      if (!p.valid) continue;
      float x, y, z;
      stencil_rt(p, j, x, y, z);
      const Taps t = make_taps(x, y, z, C, gv.D, gv.H, gv.W);
      const unsigned g2 = *(const unsigned*)((const unsigned short*)sp.dx + (row0 + r) * sp.g.Kp + col_off + j * C + 2 * cp);
      const float g0 = h2f((unsigned short)(g2 & 0xffffu)), g1 = h2f((unsigned short)(g2 >> 16));
      _Float16* base = img16 + (int64_t)p.b * gv.image_stride + 2 * cp;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        half2v v;
        v.x = (_Float16)(t.w[k] * g0); v.y = (_Float16)(t.w[k] * g1);
        __builtin_amdgcn_global_atomic_fadd_v2f16((__attribute__((address_space(1))) half2v*)(base + t.o[k]), v);
      }
    }
  }
}

__global__ __launch_bounds__(256) void k_h16_to_grad(const _Float16* __restrict__ img16, float* __restrict__ out,
                                                     int64_t n8, const float* __restrict__ scale, float unscale) {
  const float inv_s = scale[1] * unscale;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    typedef __attribute__((ext_vector_type(4))) unsigned u4;
    const u4 r = __builtin_nontemporal_load((const u4*)img16 + i);
    const unsigned w[4] = {r.x, r.y, r.z, r.w};
    float o[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      o[2 * e] = h2f((unsigned short)(w[e] & 0xffffu)) * inv_s;
      o[2 * e + 1] = h2f((unsigned short)(w[e] >> 16)) * inv_s;
    }
    typedef __attribute__((ext_vector_type(4))) float f4;
    __builtin_nontemporal_store((f4){o[0], o[1], o[2], o[3]}, (f4*)out + 2 * i);
    __builtin_nontemporal_store((f4){o[4], o[5], o[6], o[7]}, (f4*)out + 2 * i + 1);
  }
}

// Coarse levels (stencil shorter than a voxel: 16^3 and 8^3, 58 % of all tap contributions): a run of
// Morton-consecutive points touches a small box of voxels, so its contributions are summed in an LDS
// window first and the window is flushed once (8-50x fewer global atomics).  LDS float atomics are slow on
// CDNA (measured ~2.6 clk per LANE), so the window is accumulated WITHOUT atomics: thread (copy k,
// channel c) owns column c of copy k of the window and is the only one that ever touches it -- plain
// ds_read / add / ds_write, the 8 taps of a sample per batch.  A tap that the border clamp folds onto
// its neighbour (weight exactly 0) is redirected to a dummy cell so the 8 addresses of a batch never alias.
// The workgroup's 64 points are cut into runs adaptively: 8-point boxes are merged greedily while the
// merged box fits the window (8^3 level: ~32 points per run, 16^3: ~16); a box that does not fit on its
// own goes straight to memory.
constexpr int kSubPts = 8;                 // granularity of the run planner = points per record chunk
constexpr int kNSub = kScatterRows / kSubPts;
struct Run { int first, count, direct; int ox, oy, oz, nx, ny, nz, b; };

// All 7 stencil samples of a point lie within the 4 x 4 x 4 voxels around the centre's cell when the
// stencil is shorter than a voxel, and they touch only 32 of them (the "plus" the forward's
// k_gather_vox_near reads): 4 x-slots x the centre's 2x2 in (y,z), plus 2 extra y-slots and 2 extra
// z-slots over the centre's 2x2 of the other axes.  The adjoint per point is therefore ONE batch of 32
// read-add-writes with the per-axis 4-slot weight vectors of the centre / minus / plus samples:
//   X[k] = g0 wcx[k] + g1 wmx[k] + g2 wpx[k],  Y[k] = g3 wmy[k] + g4 wpy[k],  Z[k] = g5 wmz[k] + g6 wpz[k]
//   dV[kx][ky][kz] = X[kx] wcy[ky] wcz[kz] + wcx[kx] Y[ky] wcz[kz] + wcx[kx] wcy[ky] Z[kz]
struct NearRec { int o[32]; float wx[3][4], wy[3][4], wz[3][4]; int cell; int pad_[3]; };

__device__ __forceinline__ void slot_weights(const Axis& a, int cbase, float (&w)[4], bool (&used)[4]) {
  const int k0 = a.i0 - cbase;            // 0, 1 or 2
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const bool lo = k == k0, hi = (k == k0 + 1) && a.has1;
    w[k] = lo ? a.w0 : (hi ? a.w1 : 0.f);
    used[k] = used[k] || lo || hi;
  }
}

// The 32 window offsets are a function of the point's BASE CELL and the run's box alone: a slot inside the box gets its
// voxel, whether this point's samples use it or not (an unused slot receives G = 0 exactly), a slot outside the box
// (no point of the run uses it: the box covers every used tap) goes to the dummy cell.  Consecutive points of one base
// cell -- ~4 at 16^3, tens at 8^3 in Morton order -- therefore share their offsets, and the kernel sums their 32
// contributions in registers and touches the window once per cell group (round 3: the per-point read-add-write chain
// was 41 - 47 % of the kernel, profiles/r03_window_adjoint_ablation.txt).
// Built by 16 threads per point (sub = 0..15): every thread derives the nine per-axis records itself (they are cheap)
// and writes two offsets and, for sub < 9, one weight row.
__device__ __forceinline__ void build_near(const Pt& p, int W, int H, int D, const Run& r, int C, NearRec& n, int sub) {
  const float pc[3] = {p.x, p.y, p.z};
  const int S[3] = {W, H, D};
  int base[3];
#pragma unroll
  for (int ax = 0; ax < 3; ++ax) {
    const Axis c = axis_setup(pc[ax], S[ax]), m = axis_setup(pc[ax] - kDisp, S[ax]), q = axis_setup(pc[ax] + kDisp, S[ax]);
    base[ax] = c.i0 - 1;
    bool used[4] = {false, false, false, false};
    float w[3][4];
    slot_weights(c, base[ax], w[0], used);
    slot_weights(m, base[ax], w[1], used);
    slot_weights(q, base[ax], w[2], used);
    float(*dst)[4] = ax == 0 ? n.wx : (ax == 1 ? n.wy : n.wz);
#pragma unroll
    for (int v = 0; v < 3; ++v)
      if (sub == ax * 3 + v) {
#pragma unroll
        for (int k = 0; k < 4; ++k) dst[v][k] = p.valid ? w[v][k] : 0.f;
      }
  }
  const int vol = r.nx * r.ny * r.nz;
  auto cell = [&](int kx, int ky, int kz) -> int {
    const int ix = base[0] + kx - r.ox, iy = base[1] + ky - r.oy, iz = base[2] + kz - r.oz;
    if (!p.valid || ix < 0 || ix >= r.nx || iy < 0 || iy >= r.ny || iz < 0 || iz >= r.nz) return vol * C;   // dummy cell
    return ((iz * r.ny + iy) * r.nx + ix) * C;
  };
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int s = 2 * sub + e;                       // window slot 0..31 (order of the forward's shared-tap kernel)
    int kx, ky, kz;
    if (s < 16) { kx = s >> 2; ky = 1 + (s & 1); kz = 1 + ((s >> 1) & 1); }
    else if (s < 24) { kx = 1 + (s & 1); ky = (s & 4) ? 3 : 0; kz = 1 + ((s >> 1) & 1); }
    else { kx = 1 + (s & 1); ky = 1 + ((s >> 1) & 1); kz = (s & 4) ? 3 : 0; }
    n.o[s] = cell(kx, ky, kz);
  }
  if (sub == 15) n.cell = p.valid ? (base[0] + 1) | ((base[1] + 1) << 10) | ((base[2] + 1) << 20) : -1;
}

// kWinFloats: 18432 (72 KB, two workgroups per CU) for the 16^3 level, whose runs need ~100-voxel boxes;
// 9216 (36 KB, four per CU) for the 8^3 level
// fp16 image of a window level: kept at the gradient scale times kWinPkScale (headroom: a coarse voxel sums
// thousands of taps)
constexpr float kWinPkScale = 0.0625f;
// a run whose box does not fit the window, packed-half form: lanes over channel pairs, taps at the window scale
template <int C, int T>
__device__ __forceinline__ void scatter_direct_h2(const ScatterParams& sp, const ListVoxLevel& gv, int col_off,
                                                  const Pt* __restrict__ pts, int r_begin, int r_end, int64_t row0,
                                                  _Float16* __restrict__ img16) {
  constexpr int CP = C / 2, per_pass = (T / CP) > 0 ? (T / CP) : 1;
  const int tid = threadIdx.x;
  if (tid >= per_pass * CP) return;
  const int cp = tid % CP;
  for (int it = r_begin * LIST_N_STENCIL + tid / CP; it < r_end * LIST_N_STENCIL; it += per_pass) {
    const int r = it / LIST_N_STENCIL, j = it - r * LIST_N_STENCIL;
    const Pt p = pts[r];
    if (!p.valid) continue;
    float x, y, z;
    stencil_rt(p, j, x, y, z);
    const Taps t = make_taps(x, y, z, C, gv.D, gv.H, gv.W);
    const unsigned g2 = *(const unsigned*)((const unsigned short*)sp.dx + (row0 + r) * sp.g.Kp + col_off + j * C + 2 * cp);
    const float g0 = h2f((unsigned short)(g2 & 0xffffu)) * kWinPkScale, g1 = h2f((unsigned short)(g2 >> 16)) * kWinPkScale;
    _Float16* base = img16 + (int64_t)p.b * gv.image_stride + 2 * cp;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      half2v v;
      v.x = (_Float16)(t.w[k] * g0); v.y = (_Float16)(t.w[k] * g1);
      __builtin_amdgcn_global_atomic_fadd_v2f16((__attribute__((address_space(1))) half2v*)(base + t.o[k]), v);
    }
  }
}

// img16 != nullptr (fp16 operands): the window is flushed as packed halfs into a zeroed fp16 image of the level
// (scale above) and k_h16_to_grad writes the fp32 gradient afterwards -- half the atomic bytes of the flush that
// bounds these levels.
template <int C, int DXH, int kWinFloats>
__global__ __launch_bounds__((C >= 128 ? C : 128)) void k_scatter_vox_win(ScatterParams sp, ListVoxLevel gv,
                                                                             int col_off, _Float16* __restrict__ img16) {
  constexpr int T = C >= 128 ? C : 128;
  constexpr int COPIES = T / C;
  constexpr int MYP = kSubPts / COPIES;              // points per copy per chunk
  constexpr int kCap = kWinFloats / T - 1;           // window voxels (one more cell is the dummy)
  static_assert(C >= 16 && T % C == 0 && kSubPts % COPIES == 0 && T >= kScatterRows, "geometry");
  __shared__ Pt pts[kScatterRows];
  __shared__ int box[kNSub][8];                      // lo x,y,z | hi x,y,z | min image | max image
  __shared__ Run runs[kNSub];
  __shared__ int n_runs;
  __shared__ NearRec nrec[kSubPts];
  __shared__ int grp_off[T / 64][32];
  __shared__ float win[kWinFloats];
  const int tid = threadIdx.x;
  const int c = tid % C, k = tid / C;
  const int blk = xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int W = gv.W, H = gv.H, D = gv.D;
  const int64_t row0 = (int64_t)blk * kScatterRows;
  const float inv_s = sp.scale[1];
  if (tid < kNSub * 8) box[tid >> 3][tid & 7] = ((tid & 7) < 3 || (tid & 7) == 6) ? INT_MAX : INT_MIN;
  if (tid < kScatterRows) pts[tid] = load_point(sp.g, blk * kScatterRows + tid);
  __syncthreads();
  if (tid < kScatterRows && pts[tid].valid) {
    const Pt p = pts[tid];
    int* bx = box[tid / kSubPts];
    const Axis lx = axis_setup(p.x - kDisp, W), hx = axis_setup(p.x + kDisp, W);
    const Axis ly = axis_setup(p.y - kDisp, H), hy = axis_setup(p.y + kDisp, H);
    const Axis lz = axis_setup(p.z - kDisp, D), hz = axis_setup(p.z + kDisp, D);
    atomicMin(&bx[0], lx.i0); atomicMin(&bx[1], ly.i0); atomicMin(&bx[2], lz.i0);
    atomicMax(&bx[3], hx.i0 + hx.has1); atomicMax(&bx[4], hy.i0 + hy.has1); atomicMax(&bx[5], hz.i0 + hz.has1);
    atomicMin(&bx[6], p.b); atomicMax(&bx[7], p.b);
  }
  __syncthreads();
  if (tid == 0) {                                    // plan the runs
    int n = 0, i = 0;
    while (i < kNSub) {
      if (box[i][7] < box[i][6]) { ++i; continue; }  // no valid point
      int lo[3] = {box[i][0], box[i][1], box[i][2]}, hi[3] = {box[i][3], box[i][4], box[i][5]};
      const int b = box[i][6];
      Run r;
      r.first = i * kSubPts; r.count = kSubPts; r.b = b;
      const int64_t v0 = (int64_t)(hi[0] - lo[0] + 1) * (hi[1] - lo[1] + 1) * (hi[2] - lo[2] + 1);
      r.direct = (box[i][6] != box[i][7] || v0 > kCap) ? 1 : 0;
      int jn = i + 1;
      while (!r.direct && jn < kNSub && box[jn][7] >= box[jn][6] && box[jn][6] == b && box[jn][7] == b) {
        int l2[3], h2[3];
        for (int a = 0; a < 3; ++a) { l2[a] = min(lo[a], box[jn][a]); h2[a] = max(hi[a], box[jn][3 + a]); }
        const int64_t v = (int64_t)(h2[0] - l2[0] + 1) * (h2[1] - l2[1] + 1) * (h2[2] - l2[2] + 1);
        if (v > kCap) break;
        for (int a = 0; a < 3; ++a) { lo[a] = l2[a]; hi[a] = h2[a]; }
        r.count += kSubPts; ++jn;
      }
      r.ox = lo[0]; r.oy = lo[1]; r.oz = lo[2];
      r.nx = hi[0] - lo[0] + 1; r.ny = hi[1] - lo[1] + 1; r.nz = hi[2] - lo[2] + 1;
      runs[n++] = r;
      i = jn;
    }
    n_runs = n;
  }
  __syncthreads();
  float* __restrict__ gout = (float*)gv.data;
  const int nr = n_runs;
#pragma unroll 1
  for (int ri = 0; ri < nr; ++ri) {
    Run r = runs[ri];
    // (uniform: the window offsets of a cell group are then scalar arithmetic, not 32 more registers)
    r.first = __builtin_amdgcn_readfirstlane(r.first); r.count = __builtin_amdgcn_readfirstlane(r.count);
    r.direct = __builtin_amdgcn_readfirstlane(r.direct); r.b = __builtin_amdgcn_readfirstlane(r.b);
    r.ox = __builtin_amdgcn_readfirstlane(r.ox); r.oy = __builtin_amdgcn_readfirstlane(r.oy); r.oz = __builtin_amdgcn_readfirstlane(r.oz);
    r.nx = __builtin_amdgcn_readfirstlane(r.nx); r.ny = __builtin_amdgcn_readfirstlane(r.ny); r.nz = __builtin_amdgcn_readfirstlane(r.nz);
    if (r.direct) {
      if (DXH && img16) scatter_direct_h2<C, T>(sp, gv, col_off, pts, r.first, r.first + r.count, row0, img16);
      else scatter_direct<C, DXH, T>(sp, gv, col_off, pts, r.first, r.first + r.count, row0, inv_s);
      continue;
    }
    const int nx = r.nx, ny = r.ny, vol = r.nx * r.ny * r.nz;
    float* mine = win + (k * (vol + 1)) * C + c;     // my column of my copy
    for (int v = 0; v <= vol; ++v) mine[v * C] = 0.f;
    // contributions of the current base-cell group: summed in registers, added to the window once per group
    int cur_cell = -2;
    float Gacc[32];
    // the group's 32 window offsets: copied from its first point's record into a per-wave LDS row when the group
    // opens (the record itself is overwritten by the next chunk; DS operations of one wave execute in order)
    int* gro = grp_off[tid >> 6];
    auto add_group = [&]() {
#ifdef LIST_WIN_NO_RMW             // ablation (wrong results): the read-add-writes of a group collapse into one
      float gs = 0.f;
#pragma unroll
      for (int q = 0; q < 32; ++q) gs += Gacc[q];
      mine[0] += gs;
#else
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");      // (pairs with the release behind the gro[] stores)
#pragma unroll
      for (int h = 0; h < 2; ++h) {                  // two batches of 16 (addresses are distinct within the 32, or the dummy)
        float v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = mine[gro[16 * h + q]];
#pragma unroll
        for (int q = 0; q < 16; ++q) mine[gro[16 * h + q]] = v[q] + Gacc[16 * h + q];
      }
#endif
    };
#pragma unroll 1
    for (int ch = 0; ch < r.count; ch += kSubPts) {
      __syncthreads();                               // previous chunk's records are consumed
#ifndef LIST_WIN_NO_RECORDS        // ablation (wrong results): the per-point records are built once per workgroup only
      if (tid < kSubPts * 16) build_near(pts[r.first + ch + (tid >> 4)], W, H, D, r, C, nrec[tid >> 4], tid & 15);
#else
      if (tid < kSubPts * 16 && ri == 0 && ch == 0) build_near(pts[r.first + ch + (tid >> 4)], W, H, D, r, C, nrec[tid >> 4], tid & 15);
#endif
      // my points of the chunk: every dX value is requested before the first use
      float gval[MYP][LIST_N_STENCIL];
#pragma unroll
      for (int m = 0; m < MYP; ++m) {
        const int64_t o = (row0 + r.first + ch + k + COPIES * m) * sp.g.Kp + col_off + c;
#pragma unroll
        for (int j = 0; j < LIST_N_STENCIL; ++j) gval[m][j] = dx_at<DXH>(sp.dx, o + j * C);
      }
      __syncthreads();
#pragma unroll
      for (int m = 0; m < MYP; ++m) {                // (unrolled: gval[m] must stay in registers)
        const NearRec& n = nrec[k + COPIES * m];
        const float* g = gval[m];
        float X[4], Y[4], Z[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          X[q] = fmaf(g[2], n.wx[2][q], fmaf(g[1], n.wx[1][q], g[0] * n.wx[0][q]));
          Y[q] = fmaf(g[4], n.wy[2][q], g[3] * n.wy[1][q]);
          Z[q] = fmaf(g[6], n.wz[2][q], g[5] * n.wz[1][q]);
        }
        const int cell = __builtin_amdgcn_readfirstlane(n.cell);          // (the same point for the whole wave)
        if (cell != cur_cell) {
          if (cur_cell != -2) add_group();
          cur_cell = cell;
          if ((tid & 63) < 32) gro[tid & 31] = n.o[tid & 31];
          // lanes 32..63 read offsets that lanes 0..31 stored: a wavefront-scope release here (and the acquire in
          // add_group) keeps the compiler from reusing gro[] loads across the store (the hardware executes the DS
          // operations of one wave in order; the fences cost no instruction beyond an lgkmcnt wait)
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
          for (int q = 0; q < 32; ++q) Gacc[q] = 0.f;
        }
#pragma unroll
        for (int kx = 0; kx < 4; ++kx)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int ky = 1 + (q & 1), kz = 1 + (q >> 1);
            Gacc[kx * 4 + q] += fmaf(X[kx], n.wy[0][ky] * n.wz[0][kz],
                                     n.wx[0][kx] * fmaf(Y[ky], n.wz[0][kz], n.wy[0][ky] * Z[kz]));
          }
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int ka = 1 + (q & 1), kb = 1 + (q >> 1), ke = e ? 3 : 0;
            Gacc[16 + e * 4 + q] += n.wx[0][ka] * Y[ke] * n.wz[0][kb];
            Gacc[24 + e * 4 + q] += n.wx[0][ka] * n.wy[0][kb] * Z[ke];
          }
      }
    }
    if (cur_cell != -2) add_group();
    __syncthreads();
#ifdef LIST_WIN_NO_FLUSH           // ablation (wrong results): the window is never flushed
    if (sp.g.Kp < 0)
#endif
    if (DXH && img16) {
      _Float16* base16 = img16 + (int64_t)r.b * gv.image_stride;
      for (int i = tid; i < vol * (C / 2); i += T) {
        const int vox = i / (C / 2), cc = 2 * (i - vox * (C / 2));
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int kk = 0; kk < COPIES; ++kk) {
          const float2 w2 = *(const float2*)(win + (kk * (vol + 1) + vox) * C + cc);
          s0 += w2.x; s1 += w2.y;
        }
        if (s0 == 0.f && s1 == 0.f) continue;
        const int vx = vox % nx, vy = (vox / nx) % ny, vz = vox / (nx * ny);
        half2v v;
        v.x = (_Float16)(s0 * kWinPkScale); v.y = (_Float16)(s1 * kWinPkScale);
        __builtin_amdgcn_global_atomic_fadd_v2f16(
            (__attribute__((address_space(1))) half2v*)(base16 + ((int64_t)((r.oz + vz) * H + (r.oy + vy)) * W + (r.ox + vx)) * C + cc), v);
      }
    } else {
      float* base = gout + (int64_t)r.b * gv.image_stride;
      for (int i = tid; i < vol * C; i += T) {
        const int vox = i / C, cc = i - vox * C;
        float sum = 0.f;
#pragma unroll
        for (int kk = 0; kk < COPIES; ++kk) sum += win[(kk * (vol + 1) + vox) * C + cc];
        if (sum == 0.f) continue;
        const int vx = vox % nx, vy = (vox / nx) % ny, vz = vox / (nx * ny);
        atomicAdd(base + ((int64_t)((r.oz + vz) * H + (r.oy + vy)) * W + (r.ox + vx)) * C + cc, sum * inv_s);
      }
    }
    __syncthreads();                                 // the next run re-zeroes the window
  }
}

// ---- middle levels: every voxel gathers --------------------------------------------------------------------
// Where the samples are dense (32^3: ~37 contributions per voxel) the adjoint is cheaper as a GATHER:
// the (point, stencil) samples are counting-sorted by their base cell, and each voxel sums the samples
// of the <= 8 cells it is a corner of and is written ONCE with a plain store -- no atomics, no memset.
// (The direct scatter of that level moves 2.3 GB through the atomic units, 1.8 ms at the chip's rate.)
// 8 consecutive channels of a map tap / of dX as floats (one or two 16-B loads)
template <int F16>
__device__ __forceinline__ void load8(const void* __restrict__ base, int64_t i, float (&f)[8]) {
  if (F16) {
    const uint4 r = *(const uint4*)((const unsigned short*)base + i);
    f[0] = h2f((unsigned short)(r.x & 0xffff)); f[1] = h2f((unsigned short)(r.x >> 16));
    f[2] = h2f((unsigned short)(r.y & 0xffff)); f[3] = h2f((unsigned short)(r.y >> 16));
    f[4] = h2f((unsigned short)(r.z & 0xffff)); f[5] = h2f((unsigned short)(r.z >> 16));
    f[6] = h2f((unsigned short)(r.w & 0xffff)); f[7] = h2f((unsigned short)(r.w >> 16));
  } else {
    const float4 a = *(const float4*)((const float*)base + i), b = *(const float4*)((const float*)base + i + 4);
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
  }
}

struct VoxSample { int off; float fx, fy, fz; };   // dX element offset of the sample's channel 0; w1 per axis
constexpr int kScanPerBlock = 4096;

__global__ __launch_bounds__(256) void k_vs_hist(ScatterParams sp, ListVoxLevel gv, int* __restrict__ keys,
                                                 int* __restrict__ bins) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int row = t >> 3, j = t & 7;
  if (row >= sp.g.rows) return;
  int cell = -1;
  if (row < sp.g.n_valid && j < LIST_N_STENCIL) {
    const Pt p = load_point(sp.g, row);
    float x, y, z;
    stencil_rt(p, j, x, y, z);
    const Axis ax = axis_setup(x, gv.W), ay = axis_setup(y, gv.H), az = axis_setup(z, gv.D);
    cell = ((p.b * gv.D + az.i0) * gv.H + ay.i0) * gv.W + ax.i0;
    atomicAdd(&bins[cell], 1);
  }
  keys[t] = cell;
}

// in-place exclusive scan of n ints: per-block scan + block totals, scan of the totals, add back.  256-thread
// workgroups (16 consecutive ints per thread, wave scans by shuffles, one barrier): beside the backward's other streams
// a 1024-thread workgroup waits for a CU with four free wave slots per SIMD (the three launches took 0.6 ms there for
// 1 MB of counters; 0.03 ms alone).
constexpr int kScanThreads = 256, kScanPerThread = kScanPerBlock / kScanThreads;

// exclusive prefix of `sum` over the workgroup's threads; the workgroup total in `total`
__device__ __forceinline__ int block_exclusive(int sum, int* wave_tot, int& total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = sum;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int u = __shfl_up(inc, off);
    if (lane >= off) inc += u;
  }
  if (lane == 63) wave_tot[wave] = inc;
  __syncthreads();
  int before = 0;
  total = 0;
#pragma unroll
  for (int w = 0; w < kScanThreads / 64; ++w) {
    const int t = wave_tot[w];
    if (w < wave) before += t;
    total += t;
  }
  return before + inc - sum;
}

__global__ __launch_bounds__(kScanThreads) void k_scan_block(int* __restrict__ a, int n, int* __restrict__ sums) {
  __shared__ int wave_tot[kScanThreads / 64];
  const int i0 = blockIdx.x * kScanPerBlock + threadIdx.x * kScanPerThread;
  int v[kScanPerThread], sum = 0;
#pragma unroll
  for (int e = 0; e < kScanPerThread; ++e) { v[e] = (i0 + e < n) ? a[i0 + e] : 0; sum += v[e]; }
  int total;
  int run = block_exclusive(sum, wave_tot, total);
#pragma unroll
  for (int e = 0; e < kScanPerThread; ++e) { if (i0 + e < n) a[i0 + e] = run; run += v[e]; }
  if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

// nb <= 1024 block totals (kVoxGatherMaxBins / kScanPerBlock), one workgroup, 4 per thread
__global__ __launch_bounds__(kScanThreads) void k_scan_sums(int* __restrict__ sums, int nb) {
  __shared__ int wave_tot[kScanThreads / 64];
  const int i0 = threadIdx.x * 4;
  int v[4], sum = 0;
#pragma unroll
  for (int e = 0; e < 4; ++e) { v[e] = (i0 + e < nb) ? sums[i0 + e] : 0; sum += v[e]; }
  int total;
  int run = block_exclusive(sum, wave_tot, total);
#pragma unroll
  for (int e = 0; e < 4; ++e) { if (i0 + e < nb) sums[i0 + e] = run; run += v[e]; }
}

__global__ __launch_bounds__(kScanThreads) void k_scan_add(int* __restrict__ a, int n, const int* __restrict__ sums) {
  const int i0 = blockIdx.x * kScanPerBlock + threadIdx.x * kScanPerThread;
  const int add = sums[blockIdx.x];
#pragma unroll
  for (int e = 0; e < kScanPerThread; ++e)
    if (i0 + e < n) a[i0 + e] += add;
}

// bins: exclusive starts on entry, END offsets on exit
template <int C>
__global__ __launch_bounds__(256) void k_vs_scatter(ScatterParams sp, ListVoxLevel gv, const int* __restrict__ keys,
                                                    int* __restrict__ bins, VoxSample* __restrict__ recs,
                                                    int col_off) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int row = t >> 3, j = t & 7;
  if (row >= sp.g.rows) return;
  const int cell = keys[t];
  if (cell < 0) return;
  const Pt p = load_point(sp.g, row);
  float x, y, z;
  stencil_rt(p, j, x, y, z);
  const Axis ax = axis_setup(x, gv.W), ay = axis_setup(y, gv.H), az = axis_setup(z, gv.D);
  VoxSample r;
  r.off = row * sp.g.Kp + col_off + j * C;
  r.fx = ax.w1; r.fy = ay.w1; r.fz = az.w1;
  recs[atomicAdd(&bins[cell], 1)] = r;
}

// lanes over channels, 256 / C voxels per workgroup; 8 samples in flight per lane
template <int C, int DXH>
__global__ __launch_bounds__(256) void k_vs_gather(ScatterParams sp, ListVoxLevel gv, const int* __restrict__ ends,
                                                   const VoxSample* __restrict__ recs, int64_t n_vox) {
  constexpr int VPB = 256 / C, NB = 8;
  const int64_t vox = (int64_t)blockIdx.x * VPB + threadIdx.x / C;
  const int c = threadIdx.x % C;
  if (vox >= n_vox) return;
  const int W = gv.W, H = gv.H, D = gv.D;
  const int x = (int)(vox % W), y = (int)((vox / W) % H), z = (int)((vox / ((int64_t)W * H)) % D);
  const int b = (int)(vox / ((int64_t)W * H * D));
  float acc = 0.f;
  // the record ranges of the four (y, z) lines first (12 independent loads), then the samples
  int r_lo[4], r_mid[4], r_hi[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int yy = y - (q & 1), zz = z - (q >> 1);
    r_lo[q] = r_mid[q] = r_hi[q] = 0;
    if (yy < 0 || zz < 0) continue;
    // cells x-1 and x of this (y, z) line are adjacent bins: one contiguous run of records
    const int bin1 = ((b * D + zz) * H + yy) * W + x;          // cell x (corner dx = 0)
    r_mid[q] = bin1 > 0 ? ends[bin1 - 1] : 0;                  // = end of cell x-1
    r_lo[q] = x > 0 ? (bin1 > 1 ? ends[bin1 - 2] : 0) : r_mid[q];
    r_hi[q] = ends[bin1];
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int dy = q & 1, dz = q >> 1;
    const int s_lo = r_lo[q], s_mid = r_mid[q], s_hi = r_hi[q];
    for (int s = s_lo; s < s_hi; s += NB) {
      VoxSample r[NB];
#pragma unroll
      for (int u = 0; u < NB; ++u) r[u] = recs[min(s + u, s_hi - 1)];
      float g[NB];
#pragma unroll
      for (int u = 0; u < NB; ++u) g[u] = dx_at<DXH>(sp.dx, (int64_t)r[u].off + c);
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const bool dx = (s + u) < s_mid;                     // the sample lies in cell x-1: corner +1 in x
        float w = (dx ? r[u].fx : 1.f - r[u].fx) * (dy ? r[u].fy : 1.f - r[u].fy) * (dz ? r[u].fz : 1.f - r[u].fz);
        if (s + u >= s_hi) w = 0.f;
        acc = fmaf(w, g[u], acc);
      }
    }
  }
  ((float*)gv.data)[vox * C + c] = acc * sp.scale[1];
}

// The same sums with a lane per (voxel, 8 channels) instead of per (voxel, channel): a sample costs the wave one 16-B
// record load and one 16-B (fp16 dX) load per EIGHT voxel-samples instead of one of each per voxel-sample -- the kernel
// above is bound by the number of vector-memory instructions (a 2-B load per lane is a whole instruction), not by bytes.
// Every channel still adds its samples in the same order with the same weights: bit-identical results.
// Lanes of a wave belong to 64 / (C / 8) voxels with different sample counts: they idle in the longer lists' tails.
template <int C, int DXH>
__global__ __launch_bounds__(256) void k_vs_gather8(ScatterParams sp, ListVoxLevel gv, const int* __restrict__ ends,
                                                    const VoxSample* __restrict__ recs, int64_t n_vox) {
  constexpr int TPV = C / 8, VPB = 256 / TPV, NB = 4;
  const int64_t vox = (int64_t)blockIdx.x * VPB + threadIdx.x / TPV;
  const int c8 = (threadIdx.x % TPV) * 8;
  if (vox >= n_vox) return;
  const int W = gv.W, H = gv.H, D = gv.D;
  const int x = (int)(vox % W), y = (int)((vox / W) % H), z = (int)((vox / ((int64_t)W * H)) % D);
  const int b = (int)(vox / ((int64_t)W * H * D));
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  int r_lo[4], r_mid[4], r_hi[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int yy = y - (q & 1), zz = z - (q >> 1);
    r_lo[q] = r_mid[q] = r_hi[q] = 0;
    if (yy < 0 || zz < 0) continue;
    const int bin1 = ((b * D + zz) * H + yy) * W + x;          // cell x (corner dx = 0)
    r_mid[q] = bin1 > 0 ? ends[bin1 - 1] : 0;                  // = end of cell x-1
    r_lo[q] = x > 0 ? (bin1 > 1 ? ends[bin1 - 2] : 0) : r_mid[q];
    r_hi[q] = ends[bin1];
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int dy = q & 1, dz = q >> 1;
    const int s_lo = r_lo[q], s_mid = r_mid[q], s_hi = r_hi[q];
    for (int s = s_lo; s < s_hi; s += NB) {
      VoxSample r[NB];
#pragma unroll
      for (int u = 0; u < NB; ++u) r[u] = recs[min(s + u, s_hi - 1)];
      float g[NB][8];
#pragma unroll
      for (int u = 0; u < NB; ++u) load8<DXH>(sp.dx, (int64_t)r[u].off + c8, g[u]);
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const bool dx = (s + u) < s_mid;                     // the sample lies in cell x-1: corner +1 in x
        float w = (dx ? r[u].fx : 1.f - r[u].fx) * (dy ? r[u].fy : 1.f - r[u].fy) * (dz ? r[u].fz : 1.f - r[u].fz);
        if (s + u >= s_hi) w = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = fmaf(w, g[u][e], acc[e]);
      }
    }
  }
  const float inv_s = sp.scale[1];
  float* out = (float*)gv.data + vox * C + c8;
  *(float4*)out = make_float4(acc[0] * inv_s, acc[1] * inv_s, acc[2] * inv_s, acc[3] * inv_s);
  *(float4*)(out + 4) = make_float4(acc[4] * inv_s, acc[5] * inv_s, acc[6] * inv_s, acc[7] * inv_s);
}

template <int C>
static hipError_t gather_level(const ScatterParams& sp, const ListVoxLevel& gv, int col_off, int B,
                               const VoxGatherBuffers& vb, hipStream_t s) {
  const int64_t n_vox = (int64_t)B * gv.D * gv.H * gv.W;
  const int nb = (int)((n_vox + kScanPerBlock - 1) / kScanPerBlock);
  hipError_t e = hipMemsetAsync(vb.bins, 0, (size_t)n_vox * sizeof(int), s);
  if (e != hipSuccess) return e;
  const dim3 gs((unsigned)((sp.g.rows * 8 + 255) / 256));
  hipLaunchKernelGGL(k_vs_hist, gs, dim3(256), 0, s, sp, gv, vb.keys, vb.bins);
  hipLaunchKernelGGL(k_scan_block, dim3((unsigned)nb), dim3(kScanThreads), 0, s, vb.bins, (int)n_vox, vb.sums);
  hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(kScanThreads), 0, s, vb.sums, nb);
  hipLaunchKernelGGL(k_scan_add, dim3((unsigned)nb), dim3(kScanThreads), 0, s, vb.bins, (int)n_vox, vb.sums);
  hipLaunchKernelGGL(k_vs_scatter<C>, gs, dim3(256), 0, s, sp, gv, vb.keys, vb.bins, (VoxSample*)vb.recs, col_off);
  static const bool octets = [] { const char* e = getenv("LIST_VS_GATHER8"); return !(e && e[0] == '0'); }();
  if (octets && (col_off % 8) == 0 && (sp.g.Kp % 8) == 0 && (reinterpret_cast<uintptr_t>(gv.data) & 15) == 0) {
    constexpr int VPB8 = 256 / (C / 8);
    const dim3 g8((unsigned)((n_vox + VPB8 - 1) / VPB8));
    if (sp.dx_f16)
      hipLaunchKernelGGL((k_vs_gather8<C, 1>), g8, dim3(256), 0, s, sp, gv, vb.bins, (const VoxSample*)vb.recs, n_vox);
    else
      hipLaunchKernelGGL((k_vs_gather8<C, 0>), g8, dim3(256), 0, s, sp, gv, vb.bins, (const VoxSample*)vb.recs, n_vox);
    return hipGetLastError();
  }
  const dim3 gg((unsigned)((n_vox + 256 / C - 1) / (256 / C)));
  if (sp.dx_f16)
    hipLaunchKernelGGL((k_vs_gather<C, 1>), gg, dim3(256), 0, s, sp, gv, vb.bins, (const VoxSample*)vb.recs, n_vox);
  else
    hipLaunchKernelGGL((k_vs_gather<C, 0>), gg, dim3(256), 0, s, sp, gv, vb.bins, (const VoxSample*)vb.recs, n_vox);
  return hipGetLastError();
}

// scalar (C == 1) levels: one lane per (point, stencil point, tap) -- a wave is one point, its 56 taps leave in ONE
// atomic instruction in which the two lanes of an x pair are neighbours (adjacent floats: one 64-B atomic request
// for both), instead of 8 instructions of 56 unrelated addresses each (one lane per sample, a tap per instruction)
template <int DXH>
__global__ __launch_bounds__(256) void k_scatter_vox1(ScatterParams sp, ListVoxLevel gv, int col_off) {
  const int64_t total = (int64_t)sp.g.n_valid * 64;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int row = (int)(t >> 6), j = (int)(t >> 3) & 7, k = (int)t & 7;
    if (j >= LIST_N_STENCIL) continue;
    const Pt p = load_point(sp.g, row);
    float x, y, z;
    stencil_rt(p, j, x, y, z);
    const Taps tp = make_taps(x, y, z, 1, gv.D, gv.H, gv.W);
    int o = tp.o[0];
    float w = tp.w[0];
#pragma unroll
    for (int q = 1; q < 8; ++q)
      if (k == q) { o = tp.o[q]; w = tp.w[q]; }
    const float gval = dx_at<DXH>(sp.dx, (int64_t)row * sp.g.Kp + col_off + j) * sp.scale[1];
    atomicAdd((float*)gv.data + (int64_t)p.b * gv.image_stride + o, w * gval);
  }
}

template <int C>
static hipError_t scatter_level(const ScatterParams& sp, const ListVoxLevel& gv, int col_off, hipStream_t s,
                                _Float16* img16 = nullptr) {
  const dim3 grid((unsigned)(sp.g.rows / kScatterRows));
  const int big = gv.W > gv.H ? (gv.W > gv.D ? gv.W : gv.D) : (gv.H > gv.D ? gv.H : gv.D);
  const float reach = kDisp * 0.5f * (float)(big - 1);          // stencil displacement in voxels
  if constexpr (C >= 64) {
    if (reach < 0.99f) {
      constexpr int T = C >= 128 ? C : 128;
      if (reach < 0.34f) {          // 8^3: small boxes, four workgroups per CU
        // fp16, 128 channels: on the matrix cores (k_scatter_vox_box) -- alone 0.28 -> 0.12 ms; beside the other streams of
        // the forked backward the VALU kernel is the better neighbour (2 waves of 199 registers per workgroup against 4 of
        // 243: step +0.05 ms with the matrix-core form), so forked calls keep it unless LIST_SCATTER_BOX=2
        static const int box_mode = [] { const char* e = getenv("LIST_SCATTER_BOX"); return e ? atoi(e) : -1; }();
        const bool box8 = box_mode < 0 ? !sp.forked : (box_mode == 1 || box_mode == 2);
        if (box8 && (img16 || scatter_f32_diagnostic()) && scatter_box_eligible(sp, gv, col_off, kWinPkScale))
          return launch_scatter_vox_box(sp, gv, col_off, img16, s);
        if (box8 && !img16 && scatter_box_split_eligible(sp, gv, col_off))       // fp32 dX (bf16x3, bf16): bf16 hi + lo operands
          return launch_scatter_vox_box_split(sp, gv, col_off, s);
        if (sp.dx_f16) hipLaunchKernelGGL((k_scatter_vox_win<C, 1, 9216>), grid, dim3(T), 0, s, sp, gv, col_off, img16);
        else hipLaunchKernelGGL((k_scatter_vox_win<C, 0, 9216>), grid, dim3(T), 0, s, sp, gv, col_off, (_Float16*)nullptr);
      } else {
        // 16^3: the same kernel with runs of <= 128 box rows: alone 0.52 -> 0.24 ms, forked step 6.63 -> 6.29 ms
        static const int box_mode = [] { const char* e = getenv("LIST_SCATTER_BOX"); return e ? atoi(e) : -1; }();
        const bool box16 = box_mode < 0 || box_mode == 2 || box_mode == 3;
        if (box16 && (img16 || scatter_f32_diagnostic()) && scatter_box_eligible(sp, gv, col_off, kWinPkScale))
          return launch_scatter_vox_box(sp, gv, col_off, img16, s);
        if (box16 && !img16 && scatter_box_split_eligible(sp, gv, col_off))
          return launch_scatter_vox_box_split(sp, gv, col_off, s);
        if (sp.dx_f16) hipLaunchKernelGGL((k_scatter_vox_win<C, 1, 18432>), grid, dim3(T), 0, s, sp, gv, col_off, img16);
        else hipLaunchKernelGGL((k_scatter_vox_win<C, 0, 18432>), grid, dim3(T), 0, s, sp, gv, col_off, (_Float16*)nullptr);
      }
      return hipGetLastError();
    }
  }
  const int nblocks = sp.g.rows / kScatterRows;
  const int cap = sp.forked ? kDirectGridForked : kDirectGrid;
  const dim3 pgrid((unsigned)(nblocks < cap ? nblocks : cap));
  if (sp.dx_f16) hipLaunchKernelGGL((k_scatter_vox<C, 1>), pgrid, dim3(256), 0, s, sp, gv, col_off, nblocks);
  else hipLaunchKernelGGL((k_scatter_vox<C, 0>), pgrid, dim3(256), 0, s, sp, gv, col_off, nblocks);
  return hipGetLastError();
}

hipError_t launch_scatter_vox(const ScatterParams& sp, const FeatLayout& L, const ListQueryArgs& a,
                              const ListVoxLevel grad_vox[LIST_N_VOX_LEVELS], const VoxGatherBuffers& vb,
                              const ScatterStreams& st) {
  (void)a;
  const int B = (int)((sp.g.p_begin + sp.g.n_valid + sp.g.N - 1) / sp.g.N);
  int n_win_pk = 0, n_win_seen = 0;
  for (int l = 0; l < LIST_N_VOX_LEVELS; ++l) {
    const ListVoxLevel& gv = grad_vox[l];
    if (!gv.data) continue;
    hipError_t e = hipSuccess;
    // voxel-side gather where the samples are dense enough: at least kVoxGatherMinDensity samples per cell
    const int64_t n_vox = (int64_t)B * gv.D * gv.H * gv.W;
    const int big = gv.W > gv.H ? (gv.W > gv.D ? gv.W : gv.D) : (gv.H > gv.D ? gv.H : gv.D);
    const bool window_level = gv.C >= 64 && kDisp * 0.5f * (float)(big - 1) < 0.99f;
    const bool dense = vb.mode == 2 || (vb.mode == 0 && !window_level &&
                                         (double)sp.g.n_valid * LIST_N_STENCIL >= kVoxGatherMinDensity * (double)n_vox);
    if (vb.bins && dense && n_vox <= kVoxGatherMaxBins && (gv.C == 16 || gv.C == 32 || gv.C == 64 ||
                                                                           gv.C == 128 || gv.C == 256)) {
      if (bwd_knockout() & 16) continue;
      switch (gv.C) {
        case 16: e = gather_level<16>(sp, gv, L.vox_off[l], B, vb, st.gather); break;
        case 32: e = gather_level<32>(sp, gv, L.vox_off[l], B, vb, st.gather); break;
        case 64: e = gather_level<64>(sp, gv, L.vox_off[l], B, vb, st.gather); break;
        case 128: e = gather_level<128>(sp, gv, L.vox_off[l], B, vb, st.gather); break;
        default: e = gather_level<256>(sp, gv, L.vox_off[l], B, vb, st.gather); break;
      }
      if (e != hipSuccess) return e;
      continue;
    }
    if (window_level ? (bwd_knockout() & (n_win_seen == 0 ? 4 : 8)) : (bwd_knockout() & 2)) { if (window_level) ++n_win_seen; continue; }
    // the first window level shares its stream with dW0; the second one has a stream to itself (st.window2: the
    // library's own side stream when the backward is forked, else the caller's).  With both behind the (contended) 2-ms
    // dW0 the window stream ended 0.4 ms after the other two (forked backward 5.16 -> 4.98 ms with the second one behind
    // the gathers on the caller's stream); on its own stream the training step is the same with the synthetic camera
    // and 0.17 - 0.22 ms shorter with the points on the clamp, where the gathers ahead of it are the long ones
    // (profiles/r04b_window_streams_ab.txt, which also has the first level on its own stream: +1.4 ms; with the 16^3 level
    // on the matrix cores the other orders were measured again -- both levels on window2, the 16^3 level behind the
    // gathers: +0.08 / +0.4 ms, profiles/r04b_box_adjoint.txt)
    hipStream_t s = st.direct;
    if (window_level && vb.mode != 2) { s = n_win_seen == 0 ? st.window : st.window2; ++n_win_seen; }
#ifndef LIST_BWD_NO_PK_ATOMICS
    // packed-half atomics (see k_scatter_vox_h2): fp16 operands, automatic form choice, C = 32 or a non-window C = 64
    // level, and the level's fp16 image fits the scratch the caller set aside
    {
      const size_t n_elem = (size_t)B * gv.image_stride;
      if (sp.dx_f16 && vb.mode == 0 && !window_level && (gv.C == 32 || gv.C == 64) && vb.h16 &&
          n_elem * 2 <= vb.h16_bytes && n_elem % 8 == 0 && gv.image_stride == (int64_t)gv.D * gv.H * gv.W * gv.C) {
        e = hipMemsetAsync(vb.h16, 0, n_elem * 2, s);
        if (e != hipSuccess) return e;
        const int nblocks = sp.g.rows / kScatterRows;
        const int cap = sp.forked ? kDirectGridForked : kDirectGrid;
        const dim3 pgrid((unsigned)(nblocks < cap ? nblocks : cap));
        if (gv.C == 32)
          hipLaunchKernelGGL(k_scatter_vox_h2<32>, pgrid, dim3(256), 0, s, sp, gv, L.vox_off[l], nblocks, (_Float16*)vb.h16);
        else
          hipLaunchKernelGGL(k_scatter_vox_h2<64>, pgrid, dim3(256), 0, s, sp, gv, L.vox_off[l], nblocks, (_Float16*)vb.h16);
        const int64_t n8 = (int64_t)(n_elem / 8);
        const int64_t cb = (n8 + 255) / 256;
        hipLaunchKernelGGL(k_h16_to_grad, dim3((unsigned)(cb < 8192 ? cb : 8192)), dim3(256), 0, s,
                           (const _Float16*)vb.h16, (float*)gv.data, n8, sp.scale, 1.f);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
        continue;
      }
    }
#endif
#ifndef LIST_BWD_NO_PK_ATOMICS
    // window levels, fp16 operands: packed-half flush into the level's fp16 image (its own scratch slot: the window
    // levels run beside the direct ones), then one pass to fp32
    if (window_level && sp.dx_f16 && vb.mode == 0 && (gv.C == 64 || gv.C == 128 || gv.C == 256) && vb.h16w && !scatter_f32_diagnostic()) {
      const size_t n_elem = (size_t)B * gv.image_stride;
      const size_t slot = vb.h16w_bytes / 2 / 256 * 256;         // two window levels at most share the scratch
      char* img = (char*)vb.h16w + (size_t)(n_win_pk & 1) * slot;
      if (n_elem * 2 <= slot && n_elem % 8 == 0 && n_win_pk < 2 &&
          gv.image_stride == (int64_t)gv.D * gv.H * gv.W * gv.C) {
        ++n_win_pk;
        e = hipMemsetAsync(img, 0, n_elem * 2, s);
        if (e != hipSuccess) return e;
        switch (gv.C) {
          case 64: e = scatter_level<64>(sp, gv, L.vox_off[l], s, (_Float16*)img); break;
          case 128: e = scatter_level<128>(sp, gv, L.vox_off[l], s, (_Float16*)img); break;
          default: e = scatter_level<256>(sp, gv, L.vox_off[l], s, (_Float16*)img); break;
        }
        if (e != hipSuccess) return e;
        const int64_t n8 = (int64_t)(n_elem / 8);
        const int64_t cb = (n8 + 255) / 256;
        hipLaunchKernelGGL(k_h16_to_grad, dim3((unsigned)(cb < 8192 ? cb : 8192)), dim3(256), 0, s,
                           (const _Float16*)img, (float*)gv.data, n8, sp.scale, 1.f / kWinPkScale);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
        continue;
      }
    }
#endif
    e = hipMemsetAsync((void*)gv.data, 0, (size_t)B * gv.image_stride * sizeof(float), s);
    if (e != hipSuccess) return e;
    if (gv.C == 1) {
      const int nb1 = (int)(((int64_t)sp.g.n_valid * 64 + 255) / 256);
      const int cap1 = 4 * (sp.forked ? kDirectGridForked : kDirectGrid);
      const dim3 grid((unsigned)(nb1 < cap1 ? nb1 : cap1));
      if (sp.dx_f16) hipLaunchKernelGGL(k_scatter_vox1<1>, grid, dim3(256), 0, s, sp, gv, L.vox_off[l]);
      else hipLaunchKernelGGL(k_scatter_vox1<0>, grid, dim3(256), 0, s, sp, gv, L.vox_off[l]);
      e = hipGetLastError();
    } else {
      switch (gv.C) {
        case 4: e = scatter_level<4>(sp, gv, L.vox_off[l], s); break;
        case 8: e = scatter_level<8>(sp, gv, L.vox_off[l], s); break;
        case 16: e = scatter_level<16>(sp, gv, L.vox_off[l], s); break;
        case 32: e = scatter_level<32>(sp, gv, L.vox_off[l], s); break;
        case 64: e = scatter_level<64>(sp, gv, L.vox_off[l], s); break;
        case 128: e = scatter_level<128>(sp, gv, L.vox_off[l], s); break;
        case 256: e = scatter_level<256>(sp, gv, L.vox_off[l], s); break;
        default: return hipErrorInvalidValue;
      }
    }
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

// ---- perceptual map ------------------------------------------------------------------------------------------
// Per-point record of the 2-D sample, in pixel order (slot -> point): everything the map gradient
// needs, so that the per-pixel gather reads 32 B per candidate instead of re-projecting.
struct ImgRec { int x0, y0; int row; int b; float wx0, wx1, wy0, wy1; };
static_assert(sizeof(ImgRec) == 32, "ImgRec is 32 bytes");

// the projection of network/modules.py:37-47 with the intermediate values the chain rule needs
struct ProjFull { float X, Y, den, u_raw, v_raw; int pass_u, pass_v; float ix, iy; int x0, y0; float wx0, wx1, wy0, wy1; };

__device__ __forceinline__ ProjFull project_full(const float* __restrict__ T, float px, float py, float pz,
                                                 int ms, float clamp_hi) {
  ProjFull r;
  float X = px * T[0], Y = px * T[1], Z = px * T[2];
  X = fmaf(py, T[3], X); Y = fmaf(py, T[4], Y); Z = fmaf(py, T[5], Z);
  X = fmaf(pz, T[6], X); Y = fmaf(pz, T[7], Y); Z = fmaf(pz, T[8], Z);
  X = X + T[9]; Y = Y + T[10]; Z = Z + T[11];
  r.X = X; r.Y = Y; r.den = Z + 1e-8f;
  r.u_raw = __fdiv_rn(X, r.den); r.v_raw = __fdiv_rn(Y, r.den);
  r.pass_u = (r.u_raw >= 0.f && r.u_raw <= clamp_hi) ? 1 : 0;     // torch.clamp backward mask
  r.pass_v = (r.v_raw >= 0.f && r.v_raw <= clamp_hi) ? 1 : 0;
  const float u = clamp_keep_nan(r.u_raw, clamp_hi), v = clamp_keep_nan(r.v_raw, clamp_hi);
  const float half = (float)(ms - 1) * 0.5f;
  const float gx = __fdiv_rn(u - half, half), gy = __fdiv_rn(v - half, half);
  r.ix = (gx + 1.f) * half; r.iy = (gy + 1.f) * half;
  const float fx = floorf(r.ix), fy = floorf(r.iy);
  r.wx1 = r.ix - fx; r.wx0 = (fx + 1.f) - r.ix;
  r.wy1 = r.iy - fy; r.wy0 = (fy + 1.f) - r.iy;
  r.x0 = min(max((int)fx, 0), ms - 1); r.y0 = min(max((int)fy, 0), ms - 1);
  return r;
}

__device__ __forceinline__ Pt load_point_raw(const GatherParams& g, int pt) {
  Pt p;
  p.valid = true;
  const int64_t gp = g.p_begin + pt;
  p.b = (int)(gp / g.N);
  const int n = (int)(gp - (int64_t)p.b * g.N);
  const float* q = g.query + (int64_t)p.b * g.q_sb + (int64_t)n * g.q_sn;
  p.x = q[(int64_t)g.perm0 * g.q_sc] * g.scale;
  p.y = q[(int64_t)g.perm1 * g.q_sc] * g.scale;
  p.z = q[(int64_t)g.perm2 * g.q_sc] * g.scale;
  return p;
}

// slot -> (point, X row): pixel order when the forward built one, else row order
__device__ __forceinline__ void slot_point(const GatherParams& g, int slot, int& pt, int& row) {
  if (g.order_img) { pt = g.order_img[slot]; row = g.row_of[pt]; }
  else { row = slot; pt = g.order ? g.order[slot] : slot; }
}

__global__ __launch_bounds__(256) void k_img_records(GatherParams g, const float* __restrict__ trans_mat, int ms,
                                                     float clamp_hi, ImgRec* __restrict__ recs) {
  const int slot = blockIdx.x * 256 + threadIdx.x;
  if (slot >= g.n_valid) return;
  int pt, row;
  slot_point(g, slot, pt, row);
  const Pt p = load_point_raw(g, pt);
  const ProjFull pr = project_full(trans_mat + p.b * 12, p.x, p.y, p.z, ms, clamp_hi);
  ImgRec r;
  r.x0 = pr.x0; r.y0 = pr.y0; r.row = row; r.b = p.b;
  r.wx0 = pr.wx0; r.wx1 = pr.wx1; r.wy0 = pr.wy0; r.wy1 = pr.wy1;
  recs[slot] = r;
}

// One workgroup per (image, map row Y, group of 4 map columns); thread t owns channels 4t .. 4t+3 (and
// +1024 ...).  Candidates = the points of the <= 4 pixel cells (rows Y-1, Y; column groups cx-1, cx)
// whose 2x2 footprint can reach the group; each contributes w(tap) * dX[row][img_off + c].
//
// Heavy groups (round 4).  Projections that pile onto the clamp of network/modules.py:43 -- an untrained spatial
// transformer puts ~90 % of the points there -- fill a handful of border cells with tens of thousands of candidates,
// which ONE workgroup then walked one after the other (measured: this stage 0.36 -> 2.58 ms at 98 % of the points on the
// clamp, the training step 6.7 -> 8.7 ms).  A group with more than kHeavyChunk candidates is therefore cut into chunks of
// kHeavyChunk: k_img_heavy_plan numbers the chunks (one scan over the groups), k_img_heavy_partial sums each chunk in its
// own workgroup into a partial row, and the group's own workgroup adds the partial rows in chunk order instead of
// walking the candidates (no atomics: as reproducible as before).
constexpr int kImgCand = 128;
constexpr int kHeavyChunk = 1024;

__device__ __forceinline__ void img_group_cells(const int* __restrict__ bins, int slot_img, int Y, int cx, int cw,
                                                int (&beg)[4], int (&end)[4]) {
  // candidate slot ranges of the 4 cells (bins hold END offsets after the forward's scatter pass)
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int yy = Y - 1 + (k >> 1), cc = cx - 1 + (k & 1);
    beg[k] = end[k] = 0;
    if (yy < 0 || cc < 0) continue;
    const int bin = slot_img * kSortPixCells + min(yy * cw + cc, kSortPixCells - 1);
    beg[k] = bin > 0 ? bins[bin - 1] : 0;
    end[k] = bins[bin];
  }
}

// candidates [s_begin, s_end) of the record array: acc[t][e] += w(tap of map column X0 + t, row Y) * dX[row][c + e]
template <int DXH>
__device__ __forceinline__ void img_accumulate(const ScatterParams& sp, const ImgRec* __restrict__ recs, ImgRec* cand,
                                               int s_begin, int s_end, int b, int Y, int X0, int qd, int nq,
                                               int img_off, float (&acc)[4][4]) {
  for (int s0 = s_begin; s0 < s_end; s0 += kImgCand) {
    const int n = min(kImgCand, s_end - s0);
    __syncthreads();
    if ((int)threadIdx.x < n) cand[threadIdx.x] = recs[s0 + threadIdx.x];
    __syncthreads();
    for (int i = 0; i < n; ++i) {
      const ImgRec r = cand[i];
      if (r.b != b) continue;                            // a clamped bin index may mix images: never, but cheap
      const float wy = (r.y0 == Y) ? r.wy0 : ((r.y0 + 1 == Y) ? r.wy1 : 0.f);
      if ((r.y0 != Y && r.y0 + 1 != Y) || r.x0 + 1 < X0 || r.x0 > X0 + 3) continue;
      if (qd >= nq) continue;
      float gv[4];
      const int64_t o = (int64_t)r.row * sp.g.Kp + img_off + qd * 4;
      if (DXH) {
        const uint2 h = *(const uint2*)((const unsigned short*)sp.dx + o);
        gv[0] = h2f((unsigned short)(h.x & 0xffff)); gv[1] = h2f((unsigned short)(h.x >> 16));
        gv[2] = h2f((unsigned short)(h.y & 0xffff)); gv[3] = h2f((unsigned short)(h.y >> 16));
      } else {
        const float4 f = *(const float4*)((const float*)sp.dx + o);
        gv[0] = f.x; gv[1] = f.y; gv[2] = f.z; gv[3] = f.w;
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int X = X0 + t;
        const float wx = (r.x0 == X) ? r.wx0 : ((r.x0 + 1 == X) ? r.wx1 : 0.f);
        const float w = wx * wy;
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[t][e] = fmaf(w, gv[e], acc[t][e]);
      }
    }
  }
}

// chunk bookkeeping of the heavy groups (device memory in the backward workspace)
struct ImgHeavy {
  int* total;            // [1] chunks in use
  int* group_base;       // [groups] first chunk of a heavy group, -1: light
  int* chunk_group;      // [max_chunks]
  int* chunk_index;      // [max_chunks] index of the chunk inside its group
  float* partial;        // [max_chunks][4][Ct]
  int max_chunks;
};

// 1024 groups per workgroup: S_g = ceil(n_g / kHeavyChunk) for groups above kHeavyChunk candidates; a workgroup that
// holds such groups scans their S_g and takes a run of chunk slots from the global counter (*h.total, zeroed by the
// launcher; the runs need no particular order).  Without heavy groups -- the usual case -- it leaves after one barrier.
__global__ __launch_bounds__(1024) void k_img_heavy_plan(const int* __restrict__ bins, int b_first, int B, int ms,
                                                         ImgHeavy h) {
  __shared__ int part[1024];
  __shared__ int run_base;
  const int cw = (ms + 3) / 4;
  const int groups = B * ms * cw;
  const int g = blockIdx.x * 1024 + threadIdx.x;
  int S = 0;
  if (g < groups) {
    const int cx = g % cw, Y = (g / cw) % ms, b = g / (cw * ms);
    int beg[4], end[4];
    img_group_cells(bins, (b - b_first) % kSortImages, Y, cx, cw, beg, end);
    const int n = (end[0] - beg[0]) + (end[1] - beg[1]) + (end[2] - beg[2]) + (end[3] - beg[3]);
    S = n > kHeavyChunk ? (n + kHeavyChunk - 1) / kHeavyChunk : 0;
  }
  if (!__syncthreads_or(S > 0)) {
    if (g < groups) h.group_base[g] = -1;
    return;
  }
  part[threadIdx.x] = S;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const int v = (int)threadIdx.x >= off ? part[threadIdx.x - off] : 0;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  if (threadIdx.x == 1023) run_base = atomicAdd(h.total, part[1023]);
  __syncthreads();
  if (g >= groups) return;
  const int base = run_base + part[threadIdx.x] - S;
  const bool fits = S > 0 && base + S <= h.max_chunks;         // (the layout's bound holds all of them; belt and braces)
  h.group_base[g] = fits ? base : -1;
  // (every slot below min(*total, max_chunks) names a valid group, fitting or not: the partial kernel reads group_base)
  for (int i = 0; i < S && base + i < h.max_chunks; ++i) { h.chunk_group[base + i] = g; h.chunk_index[base + i] = i; }
}

// one workgroup per chunk slot: the partial sums of candidates [ci * kHeavyChunk, (ci + 1) * kHeavyChunk) of the group's
// four cells taken as one list
template <int DXH>
__global__ __launch_bounds__(256) void k_img_heavy_partial(ScatterParams sp, const ImgRec* __restrict__ recs,
                                                           const int* __restrict__ bins, int b_first, int ms, int Ct,
                                                           int img_off, ImgHeavy h) {
  __shared__ ImgRec cand[kImgCand];
  const int slot = blockIdx.x;
  if (slot >= min(*h.total, h.max_chunks)) return;
  const int g = h.chunk_group[slot], ci = h.chunk_index[slot];
  if (h.group_base[g] < 0) return;                               // (a run that did not fit: its group walks its candidates)
  const int cw = (ms + 3) / 4;
  const int cx = g % cw, Y = (g / cw) % ms, b = g / (cw * ms);
  int beg[4], end[4];
  img_group_cells(bins, (b - b_first) % kSortImages, Y, cx, cw, beg, end);
  const int nq = Ct / 4;
  const int lo = ci * kHeavyChunk, hi = lo + kHeavyChunk;           // positions in the concatenated candidate list
  for (int q0 = 0; q0 < nq; q0 += 256) {
    const int qd = q0 + threadIdx.x;
    float acc[4][4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[t][e] = 0.f;
    int pos = 0;
    for (int k = 0; k < 4; ++k) {
      const int len = end[k] - beg[k];
      const int a0 = max(lo - pos, 0), a1 = min(hi - pos, len);
      if (a0 < a1) img_accumulate<DXH>(sp, recs, cand, beg[k] + a0, beg[k] + a1, b, Y, 4 * cx, qd, nq, img_off, acc);
      pos += len;
    }
    if (qd < nq) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
        *(float4*)(h.partial + ((int64_t)slot * 4 + t) * Ct + qd * 4) = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
    }
  }
}

// GH (fp16 operands only): the map gradient is the intermediate of the adjoint resize in the same call -- written as
// halfs AT the gradient scale s (the sums of a pixel's <= 4 cells of s * dX values: saturating conversion), and
// k_img_grad_level<1> multiplies by 1 / s at its end
template <int DXH, int GH = 0>
__global__ __launch_bounds__(256) void k_img_grad_gather(ScatterParams sp, const ImgRec* __restrict__ recs,
                                                         const int* __restrict__ bins, int b_first, int ms,
                                                         int Ct, int img_off, float* __restrict__ out, ImgHeavy h) {
  __shared__ ImgRec cand[kImgCand];
  const int cw = (ms + 3) / 4;
  const int cx = blockIdx.x % cw;
  const int Y = (blockIdx.x / cw) % ms;
  const int b = blockIdx.x / (cw * ms);
  const int slot_img = (b - b_first) % kSortImages;
  const int X0 = 4 * cx;
  const float inv_s = sp.scale[1];
  const int nq = Ct / 4;                                   // channel quads

  int beg[4], end[4];
  img_group_cells(bins, slot_img, Y, cx, cw, beg, end);
  const int hbase = h.group_base ? h.group_base[blockIdx.x] : -1;
  const int n_all = (end[0] - beg[0]) + (end[1] - beg[1]) + (end[2] - beg[2]) + (end[3] - beg[3]);
  float acc[4][4];
  for (int q0 = 0; q0 < nq; q0 += 256) {
    const int qd = q0 + threadIdx.x;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[t][e] = 0.f;
    if (hbase >= 0) {
      // heavy group: its chunks were summed by k_img_heavy_partial; add the partial rows in chunk order
      const int S = (n_all + kHeavyChunk - 1) / kHeavyChunk;
      if (qd < nq)
        for (int i = 0; i < S; ++i)
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const float4 v = *(const float4*)(h.partial + ((int64_t)(hbase + i) * 4 + t) * Ct + qd * 4);
            acc[t][0] += v.x; acc[t][1] += v.y; acc[t][2] += v.z; acc[t][3] += v.w;
          }
    } else {
      for (int k = 0; k < 4; ++k)
        img_accumulate<DXH>(sp, recs, cand, beg[k], end[k], b, Y, X0, qd, nq, img_off, acc);
    }
    if (qd < nq) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int X = X0 + t;
        if (X >= ms) continue;
        const int64_t o = ((int64_t)(b * ms + Y) * ms + X) * Ct + qd * 4;
        if (GH) {
          *(uint2*)((unsigned short*)out + o) = half4(make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]));
        } else {
          float4 v = make_float4(acc[t][0] * inv_s, acc[t][1] * inv_s, acc[t][2] * inv_s, acc[t][3] * inv_s);
          *(float4*)(out + o) = v;
        }
      }
    }
  }
}

// Fallback without a pixel order (maps wider than the sort's bins, or no_sort): atomics, lanes over channels.
template <int DXH>
__global__ __launch_bounds__(256) void k_img_grad_atomic(ScatterParams sp, const ImgRec* __restrict__ recs, int ms,
                                                         int Ct, int img_off, float* __restrict__ out) {
  const int slot0 = blockIdx.x * kGatherRows;
  const float inv_s = sp.scale[1];
  for (int i = 0; i < kGatherRows; ++i) {
    const int slot = slot0 + i;
    if (slot >= sp.g.n_valid) return;
    const ImgRec r = recs[slot];
    const int x1 = min(r.x0 + 1, ms - 1), y1 = min(r.y0 + 1, ms - 1);
    float* base = out + (int64_t)r.b * ms * ms * Ct;
    float* p00 = base + (int64_t)(r.y0 * ms + r.x0) * Ct; float* p01 = base + (int64_t)(r.y0 * ms + x1) * Ct;
    float* p10 = base + (int64_t)(y1 * ms + r.x0) * Ct;   float* p11 = base + (int64_t)(y1 * ms + x1) * Ct;
    for (int c = threadIdx.x; c < Ct; c += 256) {
      const float gval = dx_at<DXH>(sp.dx, (int64_t)r.row * sp.g.Kp + img_off + c) * inv_s;
      atomicAdd(p00 + c, r.wx0 * r.wy0 * gval); atomicAdd(p01 + c, r.wx1 * r.wy0 * gval);
      atomicAdd(p10 + c, r.wx0 * r.wy1 * gval); atomicAdd(p11 + c, r.wx1 * r.wy1 * gval);
    }
  }
}

// ---- trans_mat ------------------------------------------------------------------------------------------------
// d(feature_c)/d(ix) = wy0 (m01 - m00) + wy1 (m11 - m10), d/d(iy) = wx0 (m10 - m00) + wx1 (m11 - m01)
// (grid_sampler_2d_backward; a tap outside the map counts as zero); then ix = (gx+1)*half, gx = (u-half)/half,
// u = clamp(X/den), den = Z + 1e-8, [X Y Z] = [p 1] . T.
template <int F16>
__device__ __forceinline__ float map_at(const void* __restrict__ m, int64_t i) {
  return F16 ? h2f(((const unsigned short*)m)[i]) : ((const float*)m)[i];
}

struct TransPt { int64_t o00; int sx, sy; int row, b, valid; float wx0, wx1, wy0, wy1; float X, Y, den; int pass_u, pass_v;
                 float px, py, pz; };

// grid = rows/64, block = 256: the first 64 threads project the workgroup's points (pixel order), then
// the workgroup walks them with lanes over channel octets (16-B loads of the four taps and of dX), a
// whole number of points per pass; wave shuffles + LDS reduce the two coordinate derivatives per point.
#ifndef LIST_TRANS_VGPR_ATTR
#define LIST_TRANS_VGPR_ATTR
#endif
template <int F16, int DXH>
__global__ __launch_bounds__(256) LIST_TRANS_VGPR_ATTR void k_trans_grad(ScatterParams sp, const void* __restrict__ img_map,
                                                    const float* __restrict__ trans_mat, int ms, int Ct,
                                                    float clamp_hi, int img_off, float* __restrict__ grad_T) {
  __shared__ TransPt tp[kGatherRows];
  __shared__ float s_gx[kGatherRows][8], s_gy[kGatherRows][8];     // per point: <= 4 waves x 2 halves of partials
  const int blk = xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t img_stride = (int64_t)ms * ms * Ct;
  if (threadIdx.x < kGatherRows) {
    TransPt t;
    const int slot = blk * kGatherRows + threadIdx.x;
    t.valid = slot < sp.g.n_valid ? 1 : 0;
    int pt = 0, row = 0;
    if (t.valid) slot_point(sp.g, slot, pt, row);
    const Pt p = load_point_raw(sp.g, pt);
    const ProjFull pr = project_full(trans_mat + p.b * 12, p.x, p.y, p.z, ms, clamp_hi);
    t.o00 = p.b * img_stride + (int64_t)(pr.y0 * ms + pr.x0) * Ct;
    t.sx = pr.x0 + 1 < ms ? Ct : -1;                       // -1: the tap lies outside the map (counts as zero)
    t.sy = pr.y0 + 1 < ms ? ms * Ct : -1;
    t.row = row; t.b = p.b;
    t.wx0 = pr.wx0; t.wx1 = pr.wx1; t.wy0 = pr.wy0; t.wy1 = pr.wy1;
    t.X = pr.X; t.Y = pr.Y; t.den = pr.den; t.pass_u = pr.pass_u; t.pass_v = pr.pass_v;
    t.px = p.x; t.py = p.y; t.pz = p.z;
    tp[threadIdx.x] = t;
#pragma unroll
    for (int w = 0; w < 8; ++w) { s_gx[threadIdx.x][w] = 0.f; s_gy[threadIdx.x][w] = 0.f; }
  }
  __syncthreads();
  const int lq = Ct / 8;                                   // lanes that cover one point
  if (Ct % 8 == 0 && img_off % 8 == 0 && lq >= 4 && lq <= 256 && 256 % lq == 0) {
    const int ppp = 256 / lq;                              // points per pass
    const int pp = threadIdx.x / lq, q = threadIdx.x - pp * lq;
    for (int i = 0; i < kGatherRows; i += ppp) {
      const TransPt& t = tp[i + pp];
      float gx = 0.f, gy = 0.f;
      if (t.valid) {
        float g[8], m00[8], m01[8], m10[8], m11[8];
        load8<DXH>(sp.dx, (int64_t)t.row * sp.g.Kp + img_off + q * 8, g);
        const int64_t o = t.o00 + q * 8;
        load8<F16>(img_map, o, m00);
        load8<F16>(img_map, o + (t.sx > 0 ? t.sx : 0), m01);
        load8<F16>(img_map, o + (t.sy > 0 ? t.sy : 0), m10);
        load8<F16>(img_map, o + (t.sx > 0 ? t.sx : 0) + (t.sy > 0 ? t.sy : 0), m11);
        const float kx = t.sx > 0 ? 1.f : 0.f, ky = t.sy > 0 ? 1.f : 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float a01 = m01[e] * kx, a10 = m10[e] * ky, a11 = m11[e] * kx * ky;
          gx = fmaf(g[e], t.wy0 * (a01 - m00[e]) + t.wy1 * (a11 - a10), gx);
          gy = fmaf(g[e], t.wx0 * (a10 - m00[e]) + t.wx1 * (a11 - a01), gy);
        }
      }
      // lanes of one point sit in whole waves (lq >= 64) or in aligned sub-wave groups (lq < 64).  Whole waves: the two
      // 32-lane halves keep partials of their own (row-level DPP adds only) -- with the halves combined by __shfl_xor(.,
      // 32), a ds_bpermute_b32 behind this loop's loads, ONE point's gy came out different in up to 59 of 60 forked calls
      // whenever the wave shared its CU with the weight-gradient GEMM (ds_read_b64_tr_b16 + LDS-DMA); without that one
      // instruction pair: 0 of 240 (profiles/r04b_trans_mat_interference.txt)
      if (lq >= 64) {
        for (int off = 16; off > 0; off >>= 1) { gx += __shfl_xor(gx, off); gy += __shfl_xor(gy, off); }
        if ((lane & 31) == 0) {
          const int slot = 2 * (wave % (lq / 64)) + (lane >> 5);
          s_gx[i + pp][slot] = gx; s_gy[i + pp][slot] = gy;
        }
      } else {
        for (int off = lq >> 1; off > 0; off >>= 1) { gx += __shfl_xor(gx, off); gy += __shfl_xor(gy, off); }
        if ((lane & (lq - 1)) == 0) { s_gx[i + pp][0] = gx; s_gy[i + pp][0] = gy; }
      }
    }
  } else {
    // unusual channel counts: one wave per point, scalar channels
    for (int i = wave; i < kGatherRows; i += 4) {
      const TransPt& t = tp[i];
      float gx = 0.f, gy = 0.f;
      if (t.valid)
        for (int c = lane; c < Ct; c += 64) {
          const float g = dx_at<DXH>(sp.dx, (int64_t)t.row * sp.g.Kp + img_off + c);
          const float m00 = map_at<F16>(img_map, t.o00 + c);
          const float m01 = t.sx > 0 ? map_at<F16>(img_map, t.o00 + t.sx + c) : 0.f;
          const float m10 = t.sy > 0 ? map_at<F16>(img_map, t.o00 + t.sy + c) : 0.f;
          const float m11 = (t.sx > 0 && t.sy > 0) ? map_at<F16>(img_map, t.o00 + t.sx + t.sy + c) : 0.f;
          gx = fmaf(g, t.wy0 * (m01 - m00) + t.wy1 * (m11 - m10), gx);
          gy = fmaf(g, t.wx0 * (m10 - m00) + t.wx1 * (m11 - m01), gy);
        }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) { gx += __shfl_xor(gx, off); gy += __shfl_xor(gy, off); }
      if (lane == 0) { s_gx[i][0] = gx; s_gy[i][0] = gy; }
    }
  }
  __syncthreads();
  if (threadIdx.x >= kGatherRows) return;
  // chain rule per point, then one reduction per image present in the workgroup
  float d[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) d[k] = 0.f;
  const TransPt t = tp[threadIdx.x];
  const int b = t.valid ? t.b : -1;
  if (t.valid) {
    const float* px_ = s_gx[threadIdx.x];
    const float* py_ = s_gy[threadIdx.x];
    const float gix = ((px_[0] + px_[1]) + (px_[2] + px_[3])) + ((px_[4] + px_[5]) + (px_[6] + px_[7]));
    const float giy = ((py_[0] + py_[1]) + (py_[2] + py_[3])) + ((py_[4] + py_[5]) + (py_[6] + py_[7]));
    const float half = (float)(ms - 1) * 0.5f;
    const float inv_s = sp.scale[1];
    // d/d(gx) = half * d/d(ix)  (grid_sampler unnormalize);  d/du = (1/half) * d/d(gx)
    const float du = t.pass_u ? __fdiv_rn(gix * half, half) * inv_s : 0.f;
    const float dv = t.pass_v ? __fdiv_rn(giy * half, half) * inv_s : 0.f;
    if (du != 0.f || dv != 0.f) {           // (clamped projections contribute exactly nothing, also when den == 0)
      const float dX = __fdiv_rn(du, t.den), dY = __fdiv_rn(dv, t.den);
      const float dZ = -(__fdiv_rn(du * t.X, t.den * t.den) + __fdiv_rn(dv * t.Y, t.den * t.den));
      const float h[4] = {t.px, t.py, t.pz, 1.f};
#pragma unroll
      for (int k = 0; k < 4; ++k) { d[3 * k] = h[k] * dX; d[3 * k + 1] = h[k] * dY; d[3 * k + 2] = h[k] * dZ; }
    }
  }
  // threads 0..63 are wave 0: reduce per image with shuffles
  int bmin = b < 0 ? INT_MAX : b, bmax = b;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    bmin = min(bmin, __shfl_xor(bmin, off));
    bmax = max(bmax, __shfl_xor(bmax, off));
  }
  for (int bi = bmin; bi <= bmax && bmax >= 0; ++bi) {
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      float v = (b == bi) ? d[k] : 0.f;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
      if (threadIdx.x == 0 && v != 0.f) atomicAdd(grad_T + bi * 12 + k, v);
    }
  }
}

hipError_t launch_img_grad(const ScatterParams& sp, const FeatLayout& L, const ListQueryArgs& a,
                           const int* bins_pix, int nslots, void* recs, float* grad_img_map, int map_f16,
                           float* grad_trans_mat, void* const* stage_events, hipStream_t s, void* heavy,
                           size_t heavy_bytes) {
  (void)nslots;
  const int ms = a.map_size, Ct = L.img_C;
  ImgRec* rc = (ImgRec*)recs;
  auto mark = [&](int stage) {
    if (stage_events && stage_events[stage]) (void)hipEventRecord((hipEvent_t)stage_events[stage], s);
  };
  if (grad_img_map) {
    hipLaunchKernelGGL(k_img_records, dim3((unsigned)((sp.g.n_valid + 255) / 256)), dim3(256), 0, s, sp.g,
                       a.trans_mat, ms, a.clamp_hi, rc);
    const int B = (int)((sp.g.p_begin + sp.g.n_valid + sp.g.N - 1) / sp.g.N);
    if (sp.g.order_img && bins_pix) {
      const dim3 grid((unsigned)(B * ms * ((ms + 3) / 4)));
      const int b_first = (int)(sp.g.p_begin / sp.g.N);
      // heavy groups first (scratch carved from `heavy`: [total | group_base | chunk_group | chunk_index | partial rows])
      ImgHeavy h;
      memset(&h, 0, sizeof(h));
      if (heavy && heavy_bytes >= img_heavy_bytes(sp.g.rows, Ct)) {
        char* hb = (char*)heavy;
        h.max_chunks = img_heavy_max_chunks(sp.g.rows);
        h.total = (int*)hb; hb += 256;
        h.group_base = (int*)hb; hb += (size_t)kSortImages * kSortPixCells * 4;
        h.chunk_group = (int*)hb; hb += (size_t)h.max_chunks * 4;
        h.chunk_index = (int*)hb; hb += (size_t)h.max_chunks * 4;
        hb = (char*)(((uintptr_t)hb + 255) & ~(uintptr_t)255);
        h.partial = (float*)hb;
        if ((size_t)grid.x <= (size_t)kSortImages * kSortPixCells) {
          hipError_t ez = hipMemsetAsync(h.total, 0, sizeof(int), s);
          if (ez != hipSuccess) return ez;
          hipLaunchKernelGGL(k_img_heavy_plan, dim3((grid.x + 1023) / 1024), dim3(1024), 0, s, bins_pix, b_first, B, ms, h);
          if (sp.dx_f16)
            hipLaunchKernelGGL(k_img_heavy_partial<1>, dim3((unsigned)h.max_chunks), dim3(256), 0, s, sp, rc, bins_pix, b_first,
                               ms, Ct, L.img_off, h);
          else
            hipLaunchKernelGGL(k_img_heavy_partial<0>, dim3((unsigned)h.max_chunks), dim3(256), 0, s, sp, rc, bins_pix, b_first,
                               ms, Ct, L.img_off, h);
        } else {
          h.group_base = nullptr;
        }
      }
      if (sp.dx_f16 && map_f16)
        hipLaunchKernelGGL((k_img_grad_gather<1, 1>), grid, dim3(256), 0, s, sp, rc, bins_pix, b_first,
                           ms, Ct, L.img_off, grad_img_map, h);
      else if (map_f16)
        return hipErrorInvalidValue;
      else if (sp.dx_f16)
        hipLaunchKernelGGL(k_img_grad_gather<1>, grid, dim3(256), 0, s, sp, rc, bins_pix, b_first,
                           ms, Ct, L.img_off, grad_img_map, h);
      else
        hipLaunchKernelGGL(k_img_grad_gather<0>, grid, dim3(256), 0, s, sp, rc, bins_pix, b_first,
                           ms, Ct, L.img_off, grad_img_map, h);
    } else {
      if (map_f16) return hipErrorInvalidValue;            // (list_capi.hip rejects the combination)
      hipError_t e = hipMemsetAsync(grad_img_map, 0, (size_t)B * ms * ms * Ct * sizeof(float), s);
      if (e != hipSuccess) return e;
      const dim3 grid((unsigned)((sp.g.n_valid + kGatherRows - 1) / kGatherRows));
      if (sp.dx_f16)
        hipLaunchKernelGGL(k_img_grad_atomic<1>, grid, dim3(256), 0, s, sp, rc, ms, Ct, L.img_off, grad_img_map);
      else
        hipLaunchKernelGGL(k_img_grad_atomic<0>, grid, dim3(256), 0, s, sp, rc, ms, Ct, L.img_off, grad_img_map);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  mark(LIST_BWD_IMG);
  if (grad_trans_mat) {
    hipStream_t st = s;
    const int B = (int)((sp.g.p_begin + sp.g.n_valid + sp.g.N - 1) / sp.g.N);
    hipError_t e = hipMemsetAsync(grad_trans_mat, 0, (size_t)B * 12 * sizeof(float), st);
    if (e != hipSuccess) return e;
    const dim3 grid((unsigned)(sp.g.rows / kGatherRows));
    const int f16 = a.img_dtype == LIST_MAP_F16;
    if (f16 && sp.dx_f16)
      hipLaunchKernelGGL((k_trans_grad<1, 1>), grid, dim3(256), 0, st, sp, a.img_map, a.trans_mat, ms, Ct, a.clamp_hi,
                         L.img_off, grad_trans_mat);
    else if (f16)
      hipLaunchKernelGGL((k_trans_grad<1, 0>), grid, dim3(256), 0, st, sp, a.img_map, a.trans_mat, ms, Ct, a.clamp_hi,
                         L.img_off, grad_trans_mat);
    else if (sp.dx_f16)
      hipLaunchKernelGGL((k_trans_grad<0, 1>), grid, dim3(256), 0, st, sp, a.img_map, a.trans_mat, ms, Ct, a.clamp_hi,
                         L.img_off, grad_trans_mat);
    else
      hipLaunchKernelGGL((k_trans_grad<0, 0>), grid, dim3(256), 0, st, sp, a.img_map, a.trans_mat, ms, Ct, a.clamp_hi,
                         L.img_off, grad_trans_mat);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  mark(LIST_BWD_TRANS);
  return hipSuccess;
}

// ---- adjoint of the resize (list_prep_img_maps) ----------------------------------------------------------
// out[b][c][ys][xs] = sum over map pixels (oy, ox) whose bilinear footprint contains (ys, xs) of
// wy * wx * G[b][oy][ox][coff + c], with the forward's own index/weight arithmetic (prep_kernels.hip).
// One workgroup per (image, source row ys, 64 channels), separable:
//   1. R[ox][c] = sum_oy wy(oy, ys) G[b][oy][ox][c]   threads = (4 channels, 16 x lanes), 16-B loads that
//      run 256 B along the channels; every thread keeps its <= 20 columns in registers;
//   2. out[c][xs] = sum_ox wx(ox, xs) R[ox][c]          R through LDS, per-xs tap lists in CSR form;
//   3. the [64][W] tile turns through LDS so the NCHW writes run along x.
constexpr int kAdjMaxMs = 320;             // largest supported map_size
#ifndef LIST_ADJ_CG
#define LIST_ADJ_CG 32
#endif
constexpr int kAdjCg = LIST_ADJ_CG;        // channels per workgroup (A/B: 64 = 72 KB of LDS, two workgroups per CU)
constexpr int kAdjXl = 256 / (kAdjCg / 4); // x lanes of phase 1 (a lane owns 4 channels)
constexpr int kAdjCols = (kAdjMaxMs + kAdjXl - 1) / kAdjXl;

// the forward's footprint of map index o on a source axis of S pixels
__device__ __forceinline__ void adj_footprint(int o, int S, int ms, int& i0, int& i1, float& w0, float& w1) {
  const float sc = ms > 1 ? (float)(S - 1) / (float)(ms - 1) : 0.f;
  const float f = sc * (float)o;
  i0 = min((int)f, S - 1);
  i1 = i0 + (i0 < S - 1 ? 1 : 0);
  w1 = f - (float)i0;
  w0 = 1.f - w1;
}

constexpr int kAdjTileW = 128;             // source columns per pass of phases 2-3

// All levels in ONE launch: a level's workgroups are few (896 - 1792) and each is a chain of dependent steps (tap
// lists, the row loop, two LDS phases), so five launches in a row cost ~0.11 ms each whatever their bytes; side by
// side the small levels fill the gaps of the large ones.
struct AdjLevels { ListMap2D m[LIST_N_IMG_LEVELS]; int coff[LIST_N_IMG_LEVELS], maxper[LIST_N_IMG_LEVELS],
                   wg_begin[LIST_N_IMG_LEVELS], n; };

// GH: G holds halfs at the gradient scale (k_img_grad_gather<1, 1>); out = (1 / s) * adjoint
template <int GH>
__global__ __launch_bounds__(256) void k_img_grad_level(const float* __restrict__ G, int ms, int Ct, AdjLevels lv,
                                                        const float* __restrict__ scale) {
  int l = 0;
#pragma unroll
  for (int i = 1; i < LIST_N_IMG_LEVELS; ++i)
    if (i < lv.n && (int)blockIdx.x >= lv.wg_begin[i]) l = i;
  const ListMap2D m = lv.m[l];
  const int coff = lv.coff[l], maxper = lv.maxper[l];
  const int cgroups = (m.C + kAdjCg - 1) / kAdjCg;
  const int widx = (int)blockIdx.x - lv.wg_begin[l];
  const int bx = widx / cgroups, by = widx - bx * cgroups;      // (image, source row) and 64-channel group
  extern __shared__ __attribute__((aligned(16))) float dyn[];
  // dynamic LDS: R[ms][kAdjCg] | tile[kAdjCg][kAdjTileW + 1] | wy[ms] | wx[W][maxper] | ox_first[W] | ox_cnt[W]
  float* R = dyn;
  float* tile = R + ms * kAdjCg;
  float* s_wy = tile + kAdjCg * (kAdjTileW + 1);
  float* s_wx = s_wy + ms;
  int* ox_first = (int*)(s_wx + m.W * maxper);
  int* ox_cnt = ox_first + m.W;
  __shared__ int oy_range[2];
  const int ys = bx % m.H;
  const int b = bx / m.H;
  const int c0 = by * kAdjCg;
  if (threadIdx.x == 0) { oy_range[0] = INT_MAX; oy_range[1] = INT_MIN; }
  __syncthreads();
  // map rows that touch source row ys: footprints are monotone, so they form one contiguous range
  for (int o = threadIdx.x; o < ms; o += 256) {
    int i0, i1; float w0, w1;
    adj_footprint(o, m.H, ms, i0, i1, w0, w1);
    const bool hit = i0 == ys || i1 == ys;
    s_wy[o] = hit ? (i0 == ys ? w0 : 0.f) + (i1 == ys ? w1 : 0.f) : 0.f;
    if (hit) { atomicMin(&oy_range[0], o); atomicMax(&oy_range[1], o); }
  }
  // map columns that touch source column xs: first column, count, weights
  const float scx = ms > 1 ? (float)(m.W - 1) / (float)(ms - 1) : 0.f;
  for (int xs = threadIdx.x; xs < m.W; xs += 256) {
    int o = 0;
    if (scx > 0.f) o = max(0, (int)floorf((float)(xs - 1) / scx) - 2);
    int i0, i1; float w0, w1;
    for (; o < ms; ++o) {                                  // skip columns that end before xs
      adj_footprint(o, m.W, ms, i0, i1, w0, w1);
      if (i1 >= xs) break;
    }
    ox_first[xs] = o;
    int n = 0;
    for (; o < ms && n < maxper; ++o, ++n) {
      adj_footprint(o, m.W, ms, i0, i1, w0, w1);
      if (i0 > xs) break;
      s_wx[xs * maxper + n] = (i0 == xs ? w0 : 0.f) + (i1 == xs ? w1 : 0.f);
    }
    ox_cnt[xs] = n;
  }
  __syncthreads();
  // ---- phase 1
  const int cq = threadIdx.x % (kAdjCg / 4), xl = threadIdx.x / (kAdjCg / 4);
  const int nc4 = min(kAdjCg, m.C - c0) / 4;               // channel quads that exist (C % 4 == 0)
  float4 acc[kAdjCols];
#pragma unroll
  for (int k = 0; k < kAdjCols; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (cq < nc4) {
    const int oy1 = oy_range[1];
    for (int oy = oy_range[0]; oy <= oy1; ++oy) {
      const float wy = s_wy[oy];
      const int64_t go = ((int64_t)(b * ms + oy) * ms) * Ct + coff + c0 + cq * 4;
#pragma unroll
      for (int k = 0; k < kAdjCols; ++k) {
        const int ox = xl + kAdjXl * k;
        if (ox < ms) {
          float4 v;
          if (GH) {
            const uint2 h = *(const uint2*)((const unsigned short*)G + go + (int64_t)ox * Ct);
            v = make_float4(h2f((unsigned short)(h.x & 0xffff)), h2f((unsigned short)(h.x >> 16)),
                            h2f((unsigned short)(h.y & 0xffff)), h2f((unsigned short)(h.y >> 16)));
          } else {
            v = *(const float4*)(G + go + (int64_t)ox * Ct);
          }
          acc[k].x = fmaf(wy, v.x, acc[k].x); acc[k].y = fmaf(wy, v.y, acc[k].y);
          acc[k].z = fmaf(wy, v.z, acc[k].z); acc[k].w = fmaf(wy, v.w, acc[k].w);
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < kAdjCols; ++k) {
    const int ox = xl + kAdjXl * k;
    if (ox < ms) *(float4*)(R + ox * kAdjCg + cq * 4) = acc[k];
  }
  __syncthreads();
  // ---- phases 2 and 3, kAdjTileW source columns at a time
  const int c = threadIdx.x % kAdjCg, xq = threadIdx.x / kAdjCg;
  float* out = const_cast<float*>(m.data) + (int64_t)b * m.sb + (int64_t)ys * m.sh;
  for (int x0 = 0; x0 < m.W; x0 += kAdjTileW) {
    const int wt = min(kAdjTileW, m.W - x0);
    for (int xs = xq; xs < wt; xs += 256 / kAdjCg) {
      const int first = ox_first[x0 + xs], n = ox_cnt[x0 + xs];
      const float* wx = s_wx + (x0 + xs) * maxper;
      float a = 0.f;
      for (int e = 0; e < n; ++e) a = fmaf(wx[e], R[(first + e) * kAdjCg + c], a);
      tile[c * (kAdjTileW + 1) + xs] = GH ? a * scale[1] : a;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kAdjCg * wt; i += 256) {
      const int cc = i / wt, xs = i - cc * wt;
      if (c0 + cc < m.C)
        out[(int64_t)(c0 + cc) * m.sc + (int64_t)(x0 + xs) * m.sw] = tile[cc * (kAdjTileW + 1) + xs];
    }
    __syncthreads();
  }
}

// ---- pre-pooled form: [B][C][N] <-> rows --------------------------------------------------------------------
// out[b][c][n] = (1/s) dX[row of point (b,n)][img_off + c]: the gradient of VoxelDecoder2.forward's third
// argument.  64 points x 64 channels per workgroup, turned through LDS so reads run along c, writes along n.
template <int DXH>
__global__ __launch_bounds__(256) void k_rows_to_grad(const void* __restrict__ dx, int Kp, int img_off,
                                                      const int* __restrict__ row_of, float* __restrict__ out,
                                                      int64_t sb, int64_t sc, int64_t sn, int N, int C,
                                                      const float* __restrict__ scale) {
  __shared__ float tile[64][65];
  const int nt = (N + 63) / 64;
  const int b = blockIdx.x / nt, n0 = (blockIdx.x % nt) * 64, c0 = blockIdx.y * 64;
  const int cl = threadIdx.x & 63, q = threadIdx.x >> 6;
  for (int i = q; i < 64; i += 4) {
    const int n = n0 + i;
    float v = 0.f;
    if (n < N && c0 + cl < C) {
      const int pt = b * N + n;
      const int row = row_of ? row_of[pt] : pt;
      v = dx_at<DXH>(dx, (int64_t)row * Kp + img_off + c0 + cl);
    }
    tile[i][cl] = v;
  }
  __syncthreads();
  const float inv_s = scale[1];
  for (int i = q; i < 64; i += 4) {
    const int c = c0 + i, n = n0 + cl;
    if (c < C && n < N) out[(int64_t)b * sb + (int64_t)c * sc + (int64_t)n * sn] = tile[cl][i] * inv_s;
  }
}

__global__ __launch_bounds__(256) void k_invert_order(const int* __restrict__ order, int n, int* __restrict__ row_of) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r < n) row_of[order[r]] = r;
}

hipError_t launch_rows_to_grad(const ScatterParams& sp, int img_off, int C, int B, int* row_of_scratch, float* out,
                               int64_t sb, int64_t sc, int64_t sn, hipStream_t s) {
  const int* row_of = nullptr;
  if (sp.g.order) {
    hipLaunchKernelGGL(k_invert_order, dim3((unsigned)((sp.g.n_valid + 255) / 256)), dim3(256), 0, s, sp.g.order,
                       sp.g.n_valid, row_of_scratch);
    row_of = row_of_scratch;
  }
  const dim3 grid((unsigned)(B * ((sp.g.N + 63) / 64)), (unsigned)((C + 63) / 64));
  if (sp.dx_f16)
    hipLaunchKernelGGL(k_rows_to_grad<1>, grid, dim3(256), 0, s, sp.dx, sp.g.Kp, img_off, row_of, out, sb, sc, sn, sp.g.N,
                       C, sp.scale);
  else
    hipLaunchKernelGGL(k_rows_to_grad<0>, grid, dim3(256), 0, s, sp.dx, sp.g.Kp, img_off, row_of, out, sb, sc, sn, sp.g.N,
                       C, sp.scale);
  return hipGetLastError();
}

// dx[b*N + n][c] = src[b][c][n], rows beyond B*N zero; scale = {1, 1}
__global__ __launch_bounds__(256) void k_grad_to_rows(const float* __restrict__ src, int64_t sb, int64_t sc, int64_t sn,
                                                      int N, int C, float* __restrict__ dx) {
  __shared__ float tile[64][65];
  const int nt = (N + 63) / 64;
  const int b = blockIdx.x / nt, n0 = (blockIdx.x % nt) * 64, c0 = blockIdx.y * 64;
  const int l = threadIdx.x & 63, q = threadIdx.x >> 6;
  for (int i = q; i < 64; i += 4) {
    const int c = c0 + i, n = n0 + l;
    tile[i][l] = (c < C && n < N) ? src[(int64_t)b * sb + (int64_t)c * sc + (int64_t)n * sn] : 0.f;
  }
  __syncthreads();
  for (int i = q; i < 64; i += 4) {
    const int n = n0 + i, c = c0 + l;
    if (n < N && c < C) dx[((int64_t)b * N + n) * C + c] = tile[l][i];
  }
}

__global__ void k_unit_scale(float* scale) {
  if (threadIdx.x < 4) scale[threadIdx.x] = threadIdx.x < 2 ? 1.f : 0.f;
}

hipError_t launch_grad_to_rows(const float* src, int64_t sb, int64_t sc, int64_t sn, int B, int N, int C, float* dx,
                               float* scale, hipStream_t s) {
  const dim3 grid((unsigned)(B * ((N + 63) / 64)), (unsigned)((C + 63) / 64));
  hipLaunchKernelGGL(k_grad_to_rows, grid, dim3(256), 0, s, src, sb, sc, sn, N, C, dx);
  hipLaunchKernelGGL(k_unit_scale, dim3(1), dim3(64), 0, s, scale);
  return hipGetLastError();
}

hipError_t launch_img_grad_to_levels(const float* grad_img_map, int B, int map_size, int Ct,
                                     const ListMap2D grads[LIST_N_IMG_LEVELS], hipStream_t s, int map_f16,
                                     const float* scale) {
  if (map_size > kAdjMaxMs) return hipErrorInvalidValue;
  AdjLevels lv;
  lv.n = 0;
  int coff = 0;
  int64_t wgs = 0;
  size_t lds = 0;
  for (int i = 0; i < LIST_N_IMG_LEVELS; ++i) {
    const ListMap2D& m = grads[i];
    if (m.data) {
      if (m.C % 4 || coff % 4 || Ct % 4) return hipErrorInvalidValue;
      // map columns per source column: ~2 / scale when up-sampling, at most a few when down-sampling
      const float scx = map_size > 1 ? (float)(m.W - 1) / (float)(map_size - 1) : 0.f;
      int maxper = scx > 0.f ? (int)(2.f / scx) + 4 : map_size;
      if (maxper > map_size) maxper = map_size;
      const size_t need = sizeof(float) * ((size_t)map_size * kAdjCg + kAdjCg * (size_t)(kAdjTileW + 1) + (size_t)map_size +
                                           (size_t)m.W * maxper) + sizeof(int) * 2 * (size_t)m.W;
      if (need > 150 * 1024) return hipErrorInvalidValue;
      if (need > lds) lds = need;
      const int k = lv.n++;
      lv.m[k] = m; lv.coff[k] = coff; lv.maxper[k] = maxper; lv.wg_begin[k] = (int)wgs;
      wgs += (int64_t)B * m.H * ((m.C + kAdjCg - 1) / kAdjCg);
    }
    coff += m.C;
  }
  if (lv.n == 0) return hipSuccess;
  if (wgs >= 2147483647LL) return hipErrorInvalidValue;
  if (map_f16 && !scale) return hipErrorInvalidValue;
  const void* fn = map_f16 ? (const void*)k_img_grad_level<1> : (const void*)k_img_grad_level<0>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  if (map_f16)
    hipLaunchKernelGGL(k_img_grad_level<1>, dim3((unsigned)wgs), dim3(256), lds, s, grad_img_map, map_size, Ct, lv, scale);
  else
    hipLaunchKernelGGL(k_img_grad_level<0>, dim3((unsigned)wgs), dim3(256), lds, s, grad_img_map, map_size, Ct, lv, scale);
  return hipGetLastError();
}

}  // namespace list
