// Point-parallel feature gather (HBM/L2-bound; no MFMA):
//   * k_gather_vox<C>  : 7-point stencil x trilinear (border, align_corners) sample of one
//                        channels-last voxel level  (network/modules.py:256-265)
//   * k_gather_img     : projection by trans_mat, perspective divide, clamp, bilinear sample of
//                        the channels-last 137^2 map  (network/modules.py:37-52)
//   * k_gather_tail    : scalar (C==1) voxel levels, xyz coordinates, zero padding
// All write the bf16 hi/lo feature matrix X[row][Kp] in gather order (list_common.h).
// Lanes run over channel quads (16-B loads, coalesced along C); a workgroup owns 64 points.
#include <string.h>

#include "list_common.h"
#include "point_math.h"
#include "gather_math.h"

namespace list {

// ---- exact border semantics (cold path) -----------------------------------------------------------------
// The fast reductions multiply a tap the reference SKIPS (index == size under border padding, outside the map
// under zeros padding) by its exactly-zero weight.  That is the same number unless the voxel read in its place
// holds +-inf or NaN (0 * inf = NaN where the reference adds nothing).  Such a NaN reaches every column of the
// row's fc_0 output, so fc_0's epilogue probes for NaN per 256-row tile (GemmParams.nan_tiles), k_gather_fixup
// redoes the gathers of flagged tiles with the skipped taps' VALUES forced to zero, and fc_0 runs again gated on
// the same flags (list_capi.hip).  On finite inputs the two gated launches exit at their first instruction; the
// gather kernels themselves carry no probe (one in them cost 40-90 %: it splits the tap loads into dependent
// batches).  The scalar tail re-reduces in place (one sample per lane, no measurable cost).
template <int V>
__device__ __forceinline__ bool any_nan(const float (&a)[V]) {
  bool bad = false;
#pragma unroll
  for (int c = 0; c < V; ++c) bad = bad || (a[c] != a[c]);
  return bad;
}
__device__ __forceinline__ bool wave_any(bool x) { return __builtin_amdgcn_ballot_w64(x) != 0; }
// The cold paths take their inputs through an opaque move: without it the compiler shares sub-expressions
// (unpacked taps, tap geometry) between the fast and the cold reduction and keeps them live across the fast
// one -- measured in registers: k_gather_vox<16> 36 -> 160 VGPRs, the shared-tap kernel 104 -> 256.
__device__ __forceinline__ void launder(float& x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void launder(int& x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void launder(unsigned& x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void launder(float4& r) { launder(r.x); launder(r.y); launder(r.z); launder(r.w); }
__device__ __forceinline__ void launder(uint4& r) { launder(r.x); launder(r.y); launder(r.z); launder(r.w); }

template <typename M>
__device__ __forceinline__ void tap_fma_masked(typename M::Raw r, float w, bool dead, float (&a)[M::V]) {
  float f[M::V];
  launder(r);
  M::unpack(r, f);
#pragma unroll
  for (int c = 0; c < M::V; ++c) a[c] = fmaf(dead ? 0.f : f[c], w, a[c]);
}
template <typename M>
__device__ __forceinline__ void reduce_taps_exact(const typename M::Raw (&v)[8], const Taps& t, float (&acc)[M::V]) {
#pragma unroll
  for (int c = 0; c < M::V; ++c) acc[c] = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) tap_fma_masked<M>(v[k], t.w[k], (t.dead >> k) & 1, acc);   // (tap 0 is never skipped)
}

// store V consecutive features of one row of X.  V == 8 means the values were interpolated from fp16 maps and
// cannot leave the fp16 range: no saturation step (half4_inrange)
template <int FMT, int V, bool NT = false>
__device__ __forceinline__ void store_feats(unsigned short* __restrict__ xh, unsigned short* __restrict__ xl,
                                            int64_t off, const float (&a)[V], bool valid) {
  if constexpr (FMT == FMT_FP16 && V == 8) {       // 8 halfs: one 16-B store
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    const uint2 lo = half4_inrange(valid ? make_float4(a[0], a[1], a[2], a[3]) : make_float4(0.f, 0.f, 0.f, 0.f));
    const uint2 hi = half4_inrange(valid ? make_float4(a[4], a[5], a[6], a[7]) : make_float4(0.f, 0.f, 0.f, 0.f));
    if constexpr (NT) __builtin_nontemporal_store((u32x4){lo.x, lo.y, hi.x, hi.y}, (u32x4*)(xh + off));
    else x_store<true>((u32x4){lo.x, lo.y, hi.x, hi.y}, (u32x4*)(xh + off));
    return;
  }
#pragma unroll
  for (int h = 0; h < V / 4; ++h) {
    float4 v = valid ? make_float4(a[4 * h], a[4 * h + 1], a[4 * h + 2], a[4 * h + 3])
                     : make_float4(0.f, 0.f, 0.f, 0.f);
    store_feat4<FMT>(xh, xl, off + 4 * h, v);
  }
}

template <typename M>
__device__ __forceinline__ void reduce_taps(const typename M::Raw (&v)[8], const Taps& t, float (&acc)[M::V]) {
  tap_mul<M>(v[0], t.w[0], acc);
#pragma unroll
  for (int k = 1; k < 8; ++k) tap_fma<M>(v[k], t.w[k], acc);
}

// The scalar level that shares this level's grid (C == 1, same D x H x W: the occupancy level next to the 128^3 x 16
// one) rides along (round 4): its 8 taps sit at the same voxel indices with the same weights, so the lane with sub == 0
// takes stencil point J0's sample of it and the lane with sub == 1 J1's -- 8 scalar loads beside the 16-B tap loads of
// the pair -- and k_gather_tail (56 dependent scalar taps per point in a launch of its own, 0.059 ms) is not launched.
// Same arithmetic as k_gather_tail (reduce_taps1, its NaN re-reduction, the 16-bit store of put<>): the same bits.
struct TailRide {
  const float* l0;            // image base of the scalar level (fp32, [D][H][W])
  int64_t out;                // X offset of its 7 columns in this row
  int sub, lp;                // this lane's index among the point's lanes, lanes per point
};

// Two stencil points at a time: all 16 tap loads are issued before the first use, so a wave has
// 16 KB in flight per step instead of one dependent 8-load round trip per stencil point.
template <int C, int J0, int J1, int FMT, typename M, int TAIL = 0>
__device__ __forceinline__ void gather_pair(const ListVoxLevel& lv, const void* __restrict__ base,
                                            int64_t boff, const Pt& p, unsigned short* __restrict__ xh,
                                            unsigned short* __restrict__ xl, int64_t out_off, const TailRide& tr) {
  float x0, y0, z0, x1, y1, z1;
  stencil_point<J0>(p, x0, y0, z0);
  stencil_point<J1>(p, x1, y1, z1);
  const Taps t0 = make_taps(x0, y0, z0, C, lv.D, lv.H, lv.W);
  const Taps t1 = make_taps(x1, y1, z1, C, lv.D, lv.H, lv.W);
  typename M::Raw v0[8], v1[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) v0[k] = M::load(base, boff + t0.o[k]);
  if (J1 != J0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) v1[k] = M::load(base, boff + t1.o[k]);
  }
  // the scalar level's sample of this lane: J0 (sub == 0) or J1 (the next lane; the same lane when a point has one)
  const int second = tr.lp > 1 ? 1 : 0;
  const bool take0 = TAIL && tr.sub == 0, take1 = TAIL && J1 != J0 && tr.sub == second;
  float sv[8], sw[8];
  int sdead = 0;
  if (TAIL && (take0 || take1) && !(take0 && take1)) {
    constexpr int SH = C == 4 ? 2 : C == 8 ? 3 : C == 16 ? 4 : C == 32 ? 5 : C == 64 ? 6 : C == 128 ? 7 : 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      sv[k] = tr.l0[(take0 ? t0.o[k] : t1.o[k]) >> SH];
      sw[k] = take0 ? t0.w[k] : t1.w[k];
    }
    sdead = take0 ? t0.dead : t1.dead;
  }
  float r[M::V];
  reduce_taps<M>(v0, t0, r);
  store_feats<FMT, M::V>(xh, xl, out_off + J0 * C, r, p.valid);
  if (J1 != J0) {
    reduce_taps<M>(v1, t1, r);
    store_feats<FMT, M::V>(xh, xl, out_off + J1 * C, r, p.valid);
  }
  if (TAIL && (take0 || take1) && !(take0 && take1)) {
    float a = sv[0] * sw[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) a = fmaf(sv[k], sw[k], a);
    if (a != a) {                             // (k_gather_tail: skipped taps re-reduced with their values forced to zero)
      a = sv[0] * sw[0];
#pragma unroll
      for (int k = 1; k < 8; ++k) a = fmaf(((sdead >> k) & 1) ? 0.f : sv[k], sw[k], a);
    }
    store_feat1<FMT>(xh, xl, tr.out + (take0 ? J0 : J1), p.valid ? a : 0.f);
  }
  if (TAIL && take0 && take1) {               // one lane per point (C == V): both samples, one after the other
    constexpr int SH = C == 4 ? 2 : C == 8 ? 3 : 4;
#pragma unroll
    for (int which = 0; which < 2; ++which) {
      const Taps& t = which ? t1 : t0;
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = tr.l0[t.o[k] >> SH];
      float a = v[0] * t.w[0];
#pragma unroll
      for (int k = 1; k < 8; ++k) a = fmaf(v[k], t.w[k], a);
      if (a != a) {
        a = v[0] * t.w[0];
#pragma unroll
        for (int k = 1; k < 8; ++k) a = fmaf(((t.dead >> k) & 1) ? 0.f : v[k], t.w[k], a);
      }
      store_feat1<FMT>(xh, xl, tr.out + (which ? J1 : J0), p.valid ? a : 0.f);
    }
  }
}

// grid = rows/RB, block = 256.  LP = C/V lanes share a point; a wave covers 64/LP points.  The
// workgroup's points are fetched once, in parallel, into LDS (one dependent chain order -> query
// per workgroup instead of one per iteration).
template <int C, int V> struct VoxGeom {
  static constexpr int LP = C / V;                                  // lanes per point
  static constexpr int PW = 64 / LP;                                // points per wave per iteration
  static constexpr int ITERS = (4 * PW >= kGatherRows) ? 1 : kGatherRows / (4 * PW);
  static constexpr int RB = 4 * PW * ITERS;                         // rows per workgroup: 64, 128 or 256
};

#ifndef LIST_VOX_WAVES
#define LIST_VOX_WAVES 1
#endif
// TAIL: the scalar level of the same grid, xyz and the zero padding ride along (TailRide above; k_gather_tail's work)
struct TailArgs { const float* l0; int64_t l0_stride; int l0_off, xyz_off, F; int* nan_tiles; };

template <int FMT>
__device__ __forceinline__ void write_xyz_and_pad(const GatherParams& g, const Pt& p, int64_t ro, int xyz_off, int F);

template <int C, int FMT, int F16, int TAIL = 0>
__global__ __launch_bounds__(256, LIST_VOX_WAVES) void k_gather_vox(GatherParams g, ListVoxLevel lv, int col_off, TailArgs ta) {
  using M = MapT<F16>;
  using G = VoxGeom<C, M::V>;
  __shared__ Pt pts[G::RB];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane % G::LP, psub = lane / G::LP;
  const int blk = xcd_contiguous_block(blockIdx.x, gridDim.x);
  if (threadIdx.x < G::RB) pts[threadIdx.x] = load_point(g, blk * G::RB + threadIdx.x);
  if (TAIL && ta.nan_tiles && threadIdx.x == 0 && ((blk * G::RB) % kRowTile) == 0)
    ta.nan_tiles[(blk * G::RB) / kRowTile] = 0;                    // cleared for fc_0's probe (k_gather_tail's job otherwise)
  __syncthreads();
  unsigned short* __restrict__ xh = g.x_hi;
  unsigned short* __restrict__ xl = g.x_lo;
#pragma unroll 1
  for (int it = 0; it < G::ITERS; ++it) {
    const int local = it * (4 * G::PW) + wave * G::PW + psub;
    const int row = blk * G::RB + local;
    const Pt p = pts[local];
    const int64_t boff = (int64_t)p.b * lv.image_stride + sub * M::V;
    const int64_t out_off = (int64_t)row * g.Kp + col_off + sub * M::V;
    TailRide tr;
    tr.l0 = TAIL ? ta.l0 + (int64_t)p.b * ta.l0_stride : nullptr;
    tr.out = (int64_t)row * g.Kp + ta.l0_off; tr.sub = sub; tr.lp = G::LP;
    gather_pair<C, 0, 1, FMT, M, TAIL>(lv, lv.data, boff, p, xh, xl, out_off, tr);
    gather_pair<C, 2, 3, FMT, M, TAIL>(lv, lv.data, boff, p, xh, xl, out_off, tr);
    gather_pair<C, 4, 5, FMT, M, TAIL>(lv, lv.data, boff, p, xh, xl, out_off, tr);
    gather_pair<C, 6, 6, FMT, M, TAIL>(lv, lv.data, boff, p, xh, xl, out_off, tr);
    if (TAIL && sub == (G::LP > 1 ? 1 : 0)) write_xyz_and_pad<FMT>(g, p, (int64_t)row * g.Kp, ta.xyz_off, ta.F);
  }
}

// ---- coarse levels: the 7 stencil points share their taps ---------------------------------------------
// When the displacement is shorter than one voxel (d * (size-1)/2 < 1: the 16^3 and 8^3 levels) the
// -d / +d stencil points sample either the centre's cell or the adjacent one, so all 7 samples are
// weighted sums over a "plus" of voxels: 4 x-planes x (2x2 in y,z) + 2 extra y-planes + 2 extra
// z-planes = 32 distinct taps instead of 7 x 8 = 56 (these two levels carry 58 % of all tap bytes and
// are L1-bandwidth bound).  Each axis is factored: P[k] = sum over the centre's 2x2 of the other two
// axes of plane k, then out_j = sum_k w_j[k] P[k] with the 4-entry weight vector of stencil point j
// (two non-zeros).  Same taps and weights as the reference; only the summation order differs.
__device__ __forceinline__ void window_weights(const Axis& a, int cbase, float (&w)[4]) {
  // Axis a samples planes a.i0 and a.i0+1 (if has1); cbase = index of window slot 0
  const int k0 = a.i0 - cbase;            // 0, 1 or 2
#pragma unroll
  for (int k = 0; k < 4; ++k) w[k] = (k == k0) ? a.w0 : ((k == k0 + 1 && a.has1) ? a.w1 : 0.f);
}

// plane value: sum of 4 taps with the centre 2x2 weights of the other two axes
template <typename M>
__device__ __forceinline__ void plane4(const typename M::Raw& a, const typename M::Raw& b,
                                       const typename M::Raw& c, const typename M::Raw& d,
                                       const float (&w)[4], float (&P)[M::V]) {
  tap_mul<M>(a, w[0], P); tap_fma<M>(b, w[1], P); tap_fma<M>(c, w[2], P); tap_fma<M>(d, w[3], P);
}

template <int FMT, int V>
__device__ __forceinline__ void store_wsum(const float (&P)[4][V], const float (&w)[4],
                                           unsigned short* __restrict__ xh, unsigned short* __restrict__ xl,
                                           int64_t off, bool valid) {
  float r[V];
#pragma unroll
  for (int c = 0; c < V; ++c) {
    float t = P[0][c] * w[0];
    t = fmaf(P[1][c], w[1], t); t = fmaf(P[2][c], w[2], t); t = fmaf(P[3][c], w[3], t);
    r[c] = t;
  }
  store_feats<FMT, V>(xh, xl, off, r, valid);
}

template <int C, int FMT, int F16>
__global__ __launch_bounds__(256) void k_gather_vox_near(GatherParams g, ListVoxLevel lv, int col_off) {
  using M = MapT<F16>;
  using G = VoxGeom<C, M::V>;
  using Raw = typename M::Raw;
  constexpr int V = M::V;
  __shared__ Pt pts[G::RB];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane % G::LP, psub = lane / G::LP;
  const int blk = xcd_contiguous_block(blockIdx.x, gridDim.x);
  if (threadIdx.x < G::RB) pts[threadIdx.x] = load_point(g, blk * G::RB + threadIdx.x);
  __syncthreads();
  unsigned short* __restrict__ xh = g.x_hi;
  unsigned short* __restrict__ xl = g.x_lo;
  const int W = lv.W, H = lv.H, D = lv.D;
#pragma unroll 1
  for (int it = 0; it < G::ITERS; ++it) {
    const int local = it * (4 * G::PW) + wave * G::PW + psub;
    const int row = blk * G::RB + local;
    const Pt p = pts[local];
    const void* __restrict__ base = lv.data;
    const int64_t bo = (int64_t)p.b * lv.image_stride + sub * V;
    const int64_t out_off = (int64_t)row * g.Kp + col_off + sub * V;

    const Axis cx = axis_setup(p.x, W), cy = axis_setup(p.y, H), cz = axis_setup(p.z, D);
    const Axis mx = axis_setup(p.x - kDisp, W), px = axis_setup(p.x + kDisp, W);
    const Axis my = axis_setup(p.y - kDisp, H), py = axis_setup(p.y + kDisp, H);
    const Axis mz = axis_setup(p.z - kDisp, D), pz = axis_setup(p.z + kDisp, D);
    // element offsets of the window planes (clamped; clamped-away slots always get weight 0)
    int ox[4], oy[4], oz[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      ox[k] = min(max(cx.i0 - 1 + k, 0), W - 1) * C;
      oy[k] = min(max(cy.i0 - 1 + k, 0), H - 1) * W * C;
      oz[k] = min(max(cz.i0 - 1 + k, 0), D - 1) * H * W * C;
    }
    float wcx[4], wcy[4], wcz[4], wmx[4], wpx[4], wmy[4], wpy[4], wmz[4], wpz[4];
    window_weights(cx, cx.i0 - 1, wcx); window_weights(cy, cy.i0 - 1, wcy); window_weights(cz, cz.i0 - 1, wcz);
    window_weights(mx, cx.i0 - 1, wmx); window_weights(px, cx.i0 - 1, wpx);
    window_weights(my, cy.i0 - 1, wmy); window_weights(py, cy.i0 - 1, wpy);
    window_weights(mz, cz.i0 - 1, wmz); window_weights(pz, cz.i0 - 1, wpz);
    // centre 2x2 weights of each axis pair (window slots 1 and 2)
    const float wyz[4] = {wcy[1] * wcz[1], wcy[2] * wcz[1], wcy[1] * wcz[2], wcy[2] * wcz[2]};
    const float wxz[4] = {wcx[1] * wcz[1], wcx[2] * wcz[1], wcx[1] * wcz[2], wcx[2] * wcz[2]};
    const float wxy[4] = {wcx[1] * wcy[1], wcx[2] * wcy[1], wcx[1] * wcy[2], wcx[2] * wcy[2]};

    // (Tried, round 2: three passes of 16 taps -- x, y, z axis, the 8 centre taps fetched three times -- for 168
    // instead of 218 registers, i.e. 3 instead of 2 waves per SIMD: 0.143 instead of 0.130 ms.  And the union of the
    // workgroup's 4^3 windows copied once into LDS, every tap a ds_read_b128: taken by 84 % of the workgroups of the
    // 8^3 level (median box 108 voxels = 27 KB), 0.130 ms either way.  Neither occupancy nor where the taps come
    // from moves this kernel: at ~1460 vector instructions per 4 points and 2 waves per SIMD it is bound by
    // vector-instruction issue.)
    // 4 x-planes x centre (y,z) 2x2: A[k][0..3] = (z1,y1) (z1,y2) (z2,y1) (z2,y2) of x-plane k
    Raw A[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      A[k][0] = M::load(base, bo + oz[1] + oy[1] + ox[k]);
      A[k][1] = M::load(base, bo + oz[1] + oy[2] + ox[k]);
      A[k][2] = M::load(base, bo + oz[2] + oy[1] + ox[k]);
      A[k][3] = M::load(base, bo + oz[2] + oy[2] + ox[k]);
    }
    // the two extra y-planes and z-planes over the centre 2x2 of the other two axes
    Raw By[2][4], Bz[2][4];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int ke = e ? 3 : 0;
      By[e][0] = M::load(base, bo + oz[1] + oy[ke] + ox[1]);
      By[e][1] = M::load(base, bo + oz[1] + oy[ke] + ox[2]);
      By[e][2] = M::load(base, bo + oz[2] + oy[ke] + ox[1]);
      By[e][3] = M::load(base, bo + oz[2] + oy[ke] + ox[2]);
      Bz[e][0] = M::load(base, bo + oz[ke] + oy[1] + ox[1]);
      Bz[e][1] = M::load(base, bo + oz[ke] + oy[1] + ox[2]);
      Bz[e][2] = M::load(base, bo + oz[ke] + oy[2] + ox[1]);
      Bz[e][3] = M::load(base, bo + oz[ke] + oy[2] + ox[2]);
    }
    float P[4][V];
    // x axis -> stencil points 0, 1, 2
#pragma unroll
    for (int k = 0; k < 4; ++k) plane4<M>(A[k][0], A[k][1], A[k][2], A[k][3], wyz, P[k]);
    store_wsum<FMT, V>(P, wcx, xh, xl, out_off + 0 * C, p.valid);
    store_wsum<FMT, V>(P, wmx, xh, xl, out_off + 1 * C, p.valid);
    store_wsum<FMT, V>(P, wpx, xh, xl, out_off + 2 * C, p.valid);
    // y axis: planes 1,2 come from A (x slots 1,2), planes 0,3 from By; weights (x1,z1)(x2,z1)(x1,z2)(x2,z2)
    plane4<M>(By[0][0], By[0][1], By[0][2], By[0][3], wxz, P[0]);
    plane4<M>(A[1][0], A[2][0], A[1][2], A[2][2], wxz, P[1]);
    plane4<M>(A[1][1], A[2][1], A[1][3], A[2][3], wxz, P[2]);
    plane4<M>(By[1][0], By[1][1], By[1][2], By[1][3], wxz, P[3]);
    store_wsum<FMT, V>(P, wmy, xh, xl, out_off + 3 * C, p.valid);
    store_wsum<FMT, V>(P, wpy, xh, xl, out_off + 4 * C, p.valid);
    // z axis: planes 1,2 from A, planes 0,3 from Bz; weights (x1,y1)(x2,y1)(x1,y2)(x2,y2)
    plane4<M>(Bz[0][0], Bz[0][1], Bz[0][2], Bz[0][3], wxy, P[0]);
    plane4<M>(A[1][0], A[2][0], A[1][1], A[2][1], wxy, P[1]);
    plane4<M>(A[1][2], A[2][2], A[1][3], A[2][3], wxy, P[2]);
    plane4<M>(Bz[1][0], Bz[1][1], Bz[1][2], Bz[1][3], wxy, P[3]);
    store_wsum<FMT, V>(P, wmz, xh, xl, out_off + 5 * C, p.valid);
    store_wsum<FMT, V>(P, wpz, xh, xl, out_off + 6 * C, p.valid);
  }
}

// ---- ordering of the query points ---------------------------------------------------------------------
// Random query points make every tap a cold line.  Rows are therefore processed in (image, Morton
// cell) order, a counting sort on key = image_slot * 4096 + morton(16^3 cell): consecutive rows (and
// so the workgroups resident at any moment on an XCD, see xcd_contiguous_block) sample one small
// region of the voxel maps.  The 2-D gather walks the points in a second order, (image, pixel row,
// pixel column / 4) of their projection, and writes through row_of[] into the same X rows.
// Results are independent of either order bit for bit (every point is computed on its own); the
// order inside a bin comes from atomics and is not deterministic.
__device__ __forceinline__ unsigned spread3(unsigned v) {      // 4 bits -> every third bit
  v = (v | (v << 4)) & 0x0C3u;
  v = (v | (v << 2)) & 0x249u;
  return v;
}

struct SortParams { int b_first; int pixel; const float* trans_mat; int ms; float clamp_hi; };

struct SortKeys { int morton; int pixel; };

__device__ __forceinline__ SortKeys sort_keys(const GatherParams& g, const SortParams& sp, int i) {
  const int64_t gp = g.p_begin + i;
  const int b = (int)(gp / g.N);
  const int n = (int)(gp - (int64_t)b * g.N);
  const float* q = g.query + (int64_t)b * g.q_sb + (int64_t)n * g.q_sn;
  const float px = q[(int64_t)g.perm0 * g.q_sc] * g.scale;
  const float py = q[(int64_t)g.perm1 * g.q_sc] * g.scale;
  const float pz = q[(int64_t)g.perm2 * g.q_sc] * g.scale;
  const int slot = (b - sp.b_first) % kSortImages;
  SortKeys k;
  k.pixel = 0;
  if (sp.pixel) {
    const Proj pr = project(sp.trans_mat + b * 12, px, py, pz, sp.ms, 1, sp.clamp_hi);
    const int pix = pr.o00;                               // y0 * ms + x0 (Ct = 1)
    const int y0 = pix / sp.ms, x0 = pix - y0 * sp.ms;
    k.pixel = slot * kSortPixCells + min(y0 * ((sp.ms + 3) / 4) + (x0 >> 2), kSortPixCells - 1);
  }
  const float pc[3] = {px, py, pz};
  unsigned c[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float v = (pc[a] + 1.f) * (0.5f * kSortCellsPerAxis);
    c[a] = (unsigned)fminf(fmaxf(v, 0.f), (float)(kSortCellsPerAxis - 1));
  }
  k.morton = slot * kSortCells + (int)(spread3(c[0]) | (spread3(c[1]) << 1) | (spread3(c[2]) << 2));
  return k;
}

__global__ __launch_bounds__(256) void k_zero_i32(int4* __restrict__ p, int n4) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n4) p[i] = make_int4(0, 0, 0, 0);
}

// Counter updates, one atomic per KEY per wave (round 4).  The lanes of a wave that hold the same key are found by a
// 64-step compare against every lane's key (v_readlane with a constant lane: ~4 instructions per step, no memory
// traffic) and their first lane adds the group's count in ONE atomic; every lane gets base + its rank in the group.
// Why: projections that pile onto the clamp (network/modules.py:43 -- an untrained spatial transformer puts 88 % of the
// points there, bench.py --whole-model) sent tens of thousands of atomics to a handful of pixel counters, which the
// memory side executes one after the other: the point sort took 0.74 ms instead of 0.054 ms, the whole gap between the
// module path's 2.40 ms and the bench's 1.80 ms.  Evenly spread keys cost the same as before (one atomic per lane).
// All 64 lanes must be live (no early exit before this): `active` = false lanes match nobody and add nothing.
__device__ __forceinline__ int wave_grouped_add(int* __restrict__ bins, int key, bool active, bool want_pos) {
  const int lane = threadIdx.x & 63;
  const int k = active ? key : (-1 - lane);             // real keys are >= 0
  unsigned lo = 0, hi = 0;                              // lanes that hold my key
#pragma unroll
  for (int l = 0; l < 64; ++l) {
    const int kl = __builtin_amdgcn_readlane(k, l);
    if (l < 32) lo |= (kl == k) ? (1u << l) : 0u;
    else hi |= (kl == k) ? (1u << (l - 32)) : 0u;
  }
  const unsigned below_lo = lane < 32 ? lo & ((1u << lane) - 1u) : lo;
  const unsigned below_hi = lane < 32 ? 0u : hi & ((1u << (lane - 32)) - 1u);
  const int rank = __builtin_popcount(below_lo) + __builtin_popcount(below_hi);
  const int count = __builtin_popcount(lo) + __builtin_popcount(hi);
  int base = 0;
  if (active && rank == 0) base = atomicAdd(&bins[key], count);
  if (!want_pos) return 0;
  const int first = lo ? __builtin_ctz(lo) : 32 + __builtin_ctz(hi);
  return __shfl(base, first) + rank;
}

// Both orders are built by the same three launches: bins = [Morton counters | pixel counters].
__global__ __launch_bounds__(256) void k_sort_hist(GatherParams g, SortParams sp, int* __restrict__ keys_m,
                                                   int* __restrict__ keys_p, int* __restrict__ bins_m,
                                                   int* __restrict__ bins_p) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool active = i < g.n_valid;
  SortKeys k;
  k.morton = 0; k.pixel = 0;
  if (active) {
    k = sort_keys(g, sp, i);
    keys_m[i] = k.morton;
    if (sp.pixel) keys_p[i] = k.pixel;
  }
#ifdef LIST_SORT_PLAIN_ATOMICS      // A/B: one atomic per point
  if (active) { atomicAdd(&bins_m[k.morton], 1); if (sp.pixel) atomicAdd(&bins_p[k.pixel], 1); }
#else
  wave_grouped_add(bins_m, k.morton, active, false);
  if (sp.pixel) wave_grouped_add(bins_p, k.pixel, active, false);
#endif
}

// Exclusive scan of the counters, one workgroup per (order, image slot).  The number of points of
// every slot is known on the host (points per image is fixed), so each slot's base offset arrives as
// an argument and the slots scan independently: `cells` (4096 or 8192) counters, 1024 threads.
struct SlotBase { int base[kSortImages]; };

__global__ __launch_bounds__(1024) void k_sort_scan(int* __restrict__ bins_m, int* __restrict__ bins_p,
                                                    int nslots, SlotBase sb) {
  __shared__ int part[1024];
  const bool pix = (int)blockIdx.x >= nslots;
  const int slot = pix ? blockIdx.x - nslots : blockIdx.x;
  const int cells = pix ? kSortPixCells : kSortCells;
  const int per = cells / 1024;                        // 4 or 8 consecutive counters per thread
  int* mine = (pix ? bins_p : bins_m) + (int64_t)slot * cells + threadIdx.x * per;
  int c[8];
  int sum = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) { c[i] = i < per ? mine[i] : 0; sum += c[i]; }
  part[threadIdx.x] = sum;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const int v = (int)threadIdx.x >= off ? part[threadIdx.x - off] : 0;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  int run = sb.base[slot] + part[threadIdx.x] - sum;
#pragma unroll
  for (int i = 0; i < 8; ++i)
    if (i < per) { mine[i] = run; run += c[i]; }
}

// order[pos] = i (Morton) with its inverse row_of[i] = pos, and order_img[pos'] = i (pixel order)
__global__ __launch_bounds__(256) void k_sort_scatter(int n_valid, const int* __restrict__ keys_m,
                                                      const int* __restrict__ keys_p, int* __restrict__ bins_m,
                                                      int* __restrict__ bins_p, int* __restrict__ order,
                                                      int* __restrict__ row_of, int* __restrict__ order_img) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool active = i < n_valid;
#ifdef LIST_SORT_PLAIN_ATOMICS
  if (!active) return;
  const int pos = atomicAdd(&bins_m[keys_m[i]], 1);
  order[pos] = i;
  if (order_img) {
    row_of[i] = pos;
    order_img[atomicAdd(&bins_p[keys_p[i]], 1)] = i;
  }
#else
  const int pos = wave_grouped_add(bins_m, active ? keys_m[i] : 0, active, true);
  if (active) order[pos] = i;
  if (order_img) {                                   // (uniform: a kernel argument)
    const int pp = wave_grouped_add(bins_p, active ? keys_p[i] : 0, active, true);
    if (active) { row_of[i] = pos; order_img[pp] = i; }
  }
#endif
}

hipError_t launch_sort_points(const GatherParams& g, const ListQueryArgs& a, const SortBuffers& sb,
                              hipStream_t s) {
  SortParams sp;
  sp.b_first = (int)(g.p_begin / g.N);
  const bool img = a.percep_feat == nullptr && sb.order_img != nullptr &&
                   a.map_size * ((a.map_size + 3) / 4) <= kSortPixCells;
  sp.pixel = img ? 1 : 0; sp.trans_mat = a.trans_mat; sp.ms = a.map_size; sp.clamp_hi = a.clamp_hi;
  const int64_t b_last = (g.p_begin + g.n_valid - 1) / g.N;
  const int nslots = (int)((b_last - sp.b_first + 1 < kSortImages) ? (b_last - sp.b_first + 1) : kSortImages);
  int* bins_m = sb.bins;
  int* bins_p = sb.bins + (size_t)nslots * kSortCells;
  const size_t nbins = (size_t)nslots * (kSortCells + (img ? kSortPixCells : 0));
  // points per slot: image b of the chunk goes to slot (b - b_first) % kSortImages
  SlotBase base;
  int cnt[kSortImages] = {0};
  const int64_t p_end = g.p_begin + g.n_valid;
  for (int64_t b = sp.b_first; b <= b_last; ++b) {
    const int64_t lo = b * g.N > g.p_begin ? b * g.N : g.p_begin;
    const int64_t hi = (b + 1) * g.N < p_end ? (b + 1) * g.N : p_end;
    cnt[(b - sp.b_first) % kSortImages] += (int)(hi - lo);
  }
  int run = 0;
  for (int i = 0; i < kSortImages; ++i) { base.base[i] = run; run += cnt[i]; }
  const unsigned nb = (unsigned)((g.n_valid + 255) / 256);
  GatherParams raw = g;
  raw.order = nullptr;
  int* keys_m = sb.keys;
  // the counters are cleared by a kernel of our own rather than hipMemsetAsync (one launch instead of a memset node),
  // dispatched IN STREAM ORDER: the workspace is the caller's, and a stream-ordered allocator may hand out a block
  // whose previous owner's kernel is still running -- a clear without the queue barrier (tried in round 2 for
  // ~0.01 ms) could then write zeros into that kernel's live memory
  LIST_LAUNCH(k_zero_i32, dim3((unsigned)((nbins / 4 + 255) / 256)), dim3(256), 0, s, 0, (int4*)sb.bins,
              (int)(nbins / 4));
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_sort_hist, dim3(nb), dim3(256), 0, s, raw, sp, keys_m, sb.keys2, bins_m, bins_p);
  hipLaunchKernelGGL(k_sort_scan, dim3(img ? 2 * nslots : nslots), dim3(1024), 0, s, bins_m, bins_p, nslots, base);
  hipLaunchKernelGGL(k_sort_scatter, dim3(nb), dim3(256), 0, s, g.n_valid, keys_m, sb.keys2, bins_m, bins_p,
                     sb.order, img ? sb.row_of : nullptr, img ? sb.order_img : nullptr);
  return hipGetLastError();
}

// grid = rows/64, block = 256: the workgroup walks its 64 points, lanes over channel quads.  The 64
// projections are computed once, in parallel, by the first wave (one dependent chain order -> query
// -> trans_mat per workgroup), then two points (8 x 16-B loads per lane) are in flight per step.
struct ImgPoint { Proj pr; int row; int b; int valid; };

// zeros padding as ATen's CPU kernel evaluates it: an out-of-map tap reads 0 and is still multiplied by its weight
template <typename M>
__device__ __forceinline__ void reduce_proj_exact(const typename M::Raw (&v)[4], const Proj& pr, float (&r)[M::V]) {
#pragma unroll
  for (int c = 0; c < M::V; ++c) r[c] = 0.f;
  tap_fma_masked<M>(v[0], pr.w00, pr.dead & 1, r); tap_fma_masked<M>(v[1], pr.w01, (pr.dead >> 1) & 1, r);
  tap_fma_masked<M>(v[2], pr.w10, (pr.dead >> 2) & 1, r); tap_fma_masked<M>(v[3], pr.w11, (pr.dead >> 3) & 1, r);
}

// OUT32 (projected perceptual map, list_prep_percep_proj): the map holds H1 channels per pixel and the sample is
// written as an fp32 row vector into the first bytes of the point's X row -- the perceptual block of X, which fc_0
// then leaves out of its K loop and adds in its epilogue instead.  Out-of-map taps are masked like the reference.
// `kept` (OUT32 only): the first `kept` channels of a pixel are sampled into X as without OUT32, the channels behind
// them into the row vector (list_prep_img_proj: kept encoder levels | projected sum; list_prep_percep_proj: kept = 0).
template <int FMT, int F16, int OUT32 = 0>
__global__ __launch_bounds__(256) void k_gather_img(GatherParams g, const void* __restrict__ img_map,
                                                    const float* __restrict__ trans_mat, int ms,
                                                    int Ct, float clamp_hi, int col_off, int kept) {
  using M = MapT<F16>;
  using Raw = typename M::Raw;
  __shared__ ImgPoint ipt[kGatherRows];
  const int blk = xcd_contiguous_block(blockIdx.x, gridDim.x);
  if (threadIdx.x < kGatherRows) {
    int row = blk * kGatherRows + threadIdx.x;
    Pt p;
    if (g.order_img) {          // slot -> point (pixel order) -> X row
      p.valid = row < g.n_valid;
      const int pt = p.valid ? g.order_img[row] : 0;
      const int64_t gp = g.p_begin + pt;
      p.b = (int)(gp / g.N);
      const int n = (int)(gp - (int64_t)p.b * g.N);
      const float* q = g.query + (int64_t)p.b * g.q_sb + (int64_t)n * g.q_sn;
      p.x = q[(int64_t)g.perm0 * g.q_sc] * g.scale;
      p.y = q[(int64_t)g.perm1 * g.q_sc] * g.scale;
      p.z = q[(int64_t)g.perm2 * g.q_sc] * g.scale;
      if (p.valid) row = g.row_of[pt];
    } else {
      p = load_point(g, row);
    }
    ImgPoint ip;
    ip.pr = project(trans_mat + p.b * 12, p.x, p.y, p.z, ms, Ct, clamp_hi);
    ip.row = row; ip.b = p.b; ip.valid = p.valid ? 1 : 0;
    ipt[threadIdx.x] = ip;
  }
  __syncthreads();
  unsigned short* __restrict__ xh = g.x_hi;
  unsigned short* __restrict__ xl = g.x_lo;
  const int64_t img_stride = (int64_t)ms * ms * Ct;
  constexpr int NP = 2;                 // points in flight per lane: 4 * NP 16-B loads (measured best)
  const int lq = Ct / M::V;             // lanes that cover one point
  // a step covers `span` points: the workgroup's 256 lanes over (point, channel group), NP deep
  const int ppp = lq >= 256 ? 1 : 256 / lq;                 // points per pass of the workgroup
  const int span = ppp * NP;
#pragma unroll 1
  for (int i = 0; i < kGatherRows; i += span) {
    for (int u = threadIdx.x; u < ppp * lq; u += 256) {
      const int pp = u / lq, q = u - pp * lq;
      Raw v[NP][4];
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        const int pi = min(i + k * ppp + pp, kGatherRows - 1);
        const ImgPoint& a = ipt[pi];
        const int64_t bo = a.b * img_stride + q * M::V;
        v[k][0] = M::load(img_map, bo + a.pr.o00); v[k][1] = M::load(img_map, bo + a.pr.o01);
        v[k][2] = M::load(img_map, bo + a.pr.o10); v[k][3] = M::load(img_map, bo + a.pr.o11);
      }
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        const int pi = i + k * ppp + pp;
        if (pi >= kGatherRows) continue;
        const ImgPoint& a = ipt[pi];
        float r[M::V];
        if constexpr (OUT32) {
          if (q * M::V >= kept) {
            reduce_proj_exact<M>(v[k], a.pr, r);
            float* dst = g.rowvec + (int64_t)a.row * g.rv_stride + (q * M::V - kept);
#pragma unroll
            for (int h = 0; h < M::V / 4; ++h)
              *(float4*)(dst + 4 * h) = a.valid ? make_float4(r[4 * h], r[4 * h + 1], r[4 * h + 2], r[4 * h + 3])
                                                : make_float4(0.f, 0.f, 0.f, 0.f);
            continue;
          }
        }
        tap_mul<M>(v[k][0], a.pr.w00, r); tap_fma<M>(v[k][1], a.pr.w01, r);
        tap_fma<M>(v[k][2], a.pr.w10, r); tap_fma<M>(v[k][3], a.pr.w11, r);
        // the perceptual block of a row is 2 KB of whole, line-aligned lines written once: non-temporal (round 4, two
        // interleaved pairs on one device: this kernel 0.244 -> 0.235 ms, gather group 0.850 -> 0.828, step -0.02 ms;
        // the 32 ... 256-B pieces of the voxel gathers keep the plain stores that merge in L2, list_common.h x_store)
#ifdef LIST_IMG_PLAIN_STORES
        store_feats<FMT, M::V>(xh, xl, (int64_t)a.row * g.Kp + col_off + q * M::V, r, a.valid != 0);
#else
        store_feats<FMT, M::V, true>(xh, xl, (int64_t)a.row * g.Kp + col_off + q * M::V, r, a.valid != 0);
#endif
      }
    }
  }
}

// Pre-pooled perceptual features [B,img_C,N] (VoxelDecoder2.forward's third argument) -> X.
// lane = row so reads are coalesced along N.
template <int FMT>
__global__ __launch_bounds__(256) void k_copy_percep(GatherParams g, const float* __restrict__ pf,
                                                     int64_t sb, int64_t sc, int64_t sn, int Ct,
                                                     int col_off) {
  const int row = blockIdx.x * 256 + threadIdx.x;
  if (row >= g.rows) return;
  const bool valid = row < g.n_valid;
  const int64_t gp = g.p_begin + (valid ? (g.order ? g.order[row] : row) : 0);
  const int b = (int)(gp / g.N);
  const int n = (int)(gp - (int64_t)b * g.N);
  const float* src = pf + (int64_t)b * sb + (int64_t)n * sn;
  for (int c = 0; c < Ct; c += 4) {
    float4 v = make_float4(src[(int64_t)c * sc], src[(int64_t)(c + 1) * sc],
                           src[(int64_t)(c + 2) * sc], src[(int64_t)(c + 3) * sc]);
    if (!valid) v = make_float4(0.f, 0.f, 0.f, 0.f);
    store_feat4<FMT>(g.x_hi, g.x_lo, (int64_t)row * g.Kp + col_off + c, v);
  }
}

// ---- tail: scalar voxel levels, xyz, zero padding -----------------------------------------------
struct TailLevels { ListVoxLevel lv[LIST_N_VOX_LEVELS]; int off[LIST_N_VOX_LEVELS]; int n; };

// 8 scalar taps of one trilinear sample (C == 1)
__device__ __forceinline__ float reduce_taps1(const float (&v)[8], const Taps& t) {
  float acc = v[0] * t.w[0];
#pragma unroll
  for (int k = 1; k < 8; ++k) acc = fmaf(v[k], t.w[k], acc);
  return acc;
}

template <int FMT>
__device__ __forceinline__ void put(const GatherParams& g, int64_t o, float v) {
  store_feat1<FMT>(g.x_hi, g.x_lo, o, v);
}

// 8 lanes per point: lane j < 7 samples stencil point j of every scalar level (8 taps in flight),
// lane 7 writes xyz and the zero padding.
template <int FMT>
__global__ __launch_bounds__(256) void k_gather_tail(GatherParams g, TailLevels tl, int xyz_off,
                                                     int F, int* __restrict__ nan_tiles) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int row = t >> 3, j = t & 7;
  if (row >= g.rows) return;
  if (nan_tiles && j == 7 && (row % kRowTile) == 0) nan_tiles[row / kRowTile] = 0;   // cleared for fc_0's probe
  const Pt p = load_point(g, row);
  const int64_t ro = (int64_t)row * g.Kp;
  if (j < LIST_N_STENCIL) {
    const float x = p.x + (j == 1 ? -kDisp : j == 2 ? kDisp : 0.f);
    const float y = p.y + (j == 3 ? -kDisp : j == 4 ? kDisp : 0.f);
    const float z = p.z + (j == 5 ? -kDisp : j == 6 ? kDisp : 0.f);
    for (int l = 0; l < tl.n; ++l) {
      const ListVoxLevel& lv = tl.lv[l];
      const float* __restrict__ base = (const float*)lv.data + (int64_t)p.b * lv.image_stride;
      const Taps tp = make_taps(x, y, z, 1, lv.D, lv.H, lv.W);
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = base[tp.o[k]];
      float r = reduce_taps1(v, tp);
      if (r != r) {                             // see reduce_taps_exact (per lane here: one sample per lane)
        r = v[0] * tp.w[0];
#pragma unroll
        for (int k = 1; k < 8; ++k) r = fmaf(((tp.dead >> k) & 1) ? 0.f : v[k], tp.w[k], r);
      }
      put<FMT>(g, ro + tl.off[l] + j, p.valid ? r : 0.f);
    }
    return;
  }
  write_xyz_and_pad<FMT>(g, p, ro, xyz_off, F);
}

template <int FMT>
__device__ __forceinline__ void write_xyz_and_pad(const GatherParams& g, const Pt& p, int64_t ro, int xyz_off, int F) {
  put<FMT>(g, ro + xyz_off + 0, p.valid ? p.x : 0.f);     // p_features, modules.py:257
  put<FMT>(g, ro + xyz_off + 1, p.valid ? p.y : 0.f);
  put<FMT>(g, ro + xyz_off + 2, p.valid ? p.z : 0.f);
  // zero padding up to Kp (a multiple of 64, rows are 128-B aligned): widen the stores as the
  // alignment allows -- 6 stores instead of 38 for F = 3610 (uniform control flow: F, Kp are uniform)
  // (split formats: hi and lo halfs interleaved in 64-B blocks, list_common.h xi_off; aligned runs of <= 8 stay inside a block)
  auto zero_pad = [&](unsigned short* __restrict__ xp) {
    auto at = [&](int k) -> unsigned short* { return xp + (FMT == FMT_FP16 ? ro + k : xi_off(ro + k)); };
    int k = F;
    if ((k & 1) && k < g.Kp) { *at(k) = 0; k += 1; }
    if ((k & 2) && k + 2 <= g.Kp) { *(unsigned*)at(k) = 0u; k += 2; }
    if ((k & 4) && k + 4 <= g.Kp) { *(uint2*)at(k) = make_uint2(0u, 0u); k += 4; }
    for (; k + 8 <= g.Kp; k += 8) *(uint4*)at(k) = make_uint4(0u, 0u, 0u, 0u);
  };
  zero_pad(g.x_hi);
  if (FMT == FMT_BF16_SPLIT) zero_pad(g.x_hi + kXiLo);
}

// ---- exact redo of flagged row tiles (cold) -----------------------------------------------------------------
// grid = rows / 64; a workgroup whose 256-row tile is not flagged exits at once.  Channel counts are runtime
// values here (one kernel for every level), every sample reloads its own 8 / 4 taps: slow and simple.
struct FixupArgs {
  ListVoxLevel lv[LIST_N_VOX_LEVELS]; int off[LIST_N_VOX_LEVELS]; int n;     // vector voxel levels
  const void* img_map; const float* trans_mat; int img_f16, ms, Ct, img_off; float clamp_hi;   // img_map == NULL: none
  int Cs;                                                                      // channels per pixel of img_map (Ct of them are sampled)
  const int* tile_flags;                                                       // [rows / 256]
};

template <int FMT, typename M>
__device__ __forceinline__ void fixup_level(const GatherParams& g, const ListVoxLevel& lv, int col_off, int blk,
                                            const Pt* pts) {
  const int C = lv.C, lp = C / M::V;
  for (int item = threadIdx.x; item < kGatherRows * lp; item += 256) {
    const int local = item / lp, sub = item - local * lp;
    const Pt p = pts[local];
    const int64_t boff = (int64_t)p.b * lv.image_stride + sub * M::V;
    const int64_t out_off = (int64_t)(blk * kGatherRows + local) * g.Kp + col_off + sub * M::V;
#pragma unroll 1
    for (int j = 0; j < LIST_N_STENCIL; ++j) {
      const float x = p.x + (j == 1 ? -kDisp : j == 2 ? kDisp : 0.f);
      const float y = p.y + (j == 3 ? -kDisp : j == 4 ? kDisp : 0.f);
      const float z = p.z + (j == 5 ? -kDisp : j == 6 ? kDisp : 0.f);
      const Taps t = make_taps(x, y, z, C, lv.D, lv.H, lv.W);
      typename M::Raw v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = M::load(lv.data, boff + t.o[k]);
      float r[M::V];
      reduce_taps_exact<M>(v, t, r);
      store_feats<FMT, M::V>(g.x_hi, g.x_lo, out_off + j * C, r, p.valid);
    }
  }
}

template <int FMT, typename M>
__device__ __forceinline__ void fixup_img(const GatherParams& g, const FixupArgs& fa, int blk, const Pt* pts) {
  // fa.Ct sampled channels go to X; the fa.Cs - fa.Ct channels behind them (list_prep_img_proj's projected sum: rowvec
  // set) to the row vector -- the fused fc_0 samples those itself and writes no row vector, the gated re-run reads one
  const int lq = (g.rowvec ? fa.Cs : fa.Ct) / M::V;
  const int64_t img_stride = (int64_t)fa.ms * fa.ms * fa.Cs;
  for (int item = threadIdx.x; item < kGatherRows * lq; item += 256) {
    const int local = item / lq, q = item - local * lq;
    const Pt p = pts[local];
    const Proj pr = project(fa.trans_mat + p.b * 12, p.x, p.y, p.z, fa.ms, fa.Cs, fa.clamp_hi);
    const int64_t bo = p.b * img_stride + q * M::V;
    const typename M::Raw v[4] = {M::load(fa.img_map, bo + pr.o00), M::load(fa.img_map, bo + pr.o01),
                                  M::load(fa.img_map, bo + pr.o10), M::load(fa.img_map, bo + pr.o11)};
    float r[M::V];
    reduce_proj_exact<M>(v, pr, r);
    if (q * M::V >= fa.Ct) {
      float* dst = g.rowvec + (int64_t)(blk * kGatherRows + local) * g.rv_stride + (q * M::V - fa.Ct);
#pragma unroll
      for (int h = 0; h < M::V / 4; ++h)
        *(float4*)(dst + 4 * h) = p.valid ? make_float4(r[4 * h], r[4 * h + 1], r[4 * h + 2], r[4 * h + 3])
                                          : make_float4(0.f, 0.f, 0.f, 0.f);
      continue;
    }
    store_feats<FMT, M::V>(g.x_hi, g.x_lo, (int64_t)(blk * kGatherRows + local) * g.Kp + fa.img_off + q * M::V, r,
                           p.valid);
  }
}

template <int FMT>
__global__ __launch_bounds__(256) void k_gather_fixup(GatherParams g, FixupArgs fa) {
  __shared__ Pt pts[kGatherRows];
  const int blk = blockIdx.x;
  if (fa.tile_flags[blk / (kRowTile / kGatherRows)] == 0) return;
  if (threadIdx.x < kGatherRows) pts[threadIdx.x] = load_point(g, blk * kGatherRows + threadIdx.x);
  __syncthreads();
  for (int l = 0; l < fa.n; ++l) {
    if (fa.lv[l].dtype == LIST_MAP_F16) fixup_level<FMT, MapT<1>>(g, fa.lv[l], fa.off[l], blk, pts);
    else fixup_level<FMT, MapT<0>>(g, fa.lv[l], fa.off[l], blk, pts);
  }
  if (fa.img_map) {
    if (fa.img_f16) fixup_img<FMT, MapT<1>>(g, fa, blk, pts);
    else fixup_img<FMT, MapT<0>>(g, fa, blk, pts);
  }
}

hipError_t launch_gather_fixup(const GatherParams& g, const FeatLayout& L, const ListQueryArgs& a,
                               const int* tile_flags, hipStream_t s) {
  FixupArgs fa;
  fa.n = 0;
  for (int l = 0; l < LIST_N_VOX_LEVELS; ++l)
    if (a.vox[l].C != 1) { fa.lv[fa.n] = a.vox[l]; fa.off[fa.n] = L.vox_off[l]; ++fa.n; }
  fa.img_map = (a.percep_feat || a.percep_proj) ? nullptr : a.img_map;     // (the projected sample is masked already)
  fa.trans_mat = a.trans_mat; fa.img_f16 = a.img_dtype == LIST_MAP_F16; fa.ms = a.map_size; fa.Ct = L.img_C;
  fa.Cs = L.img_C;
  if (a.img_proj) {               // only the kept levels' columns are part of X (the projected sample is masked already)
    fa.Ct = a.img_kept_C; fa.Cs = a.img_kept_C + a.H1;
  }
  fa.img_off = L.img_off; fa.clamp_hi = a.clamp_hi; fa.tile_flags = tile_flags;
  if (g.fmt == FMT_FP16)
    hipLaunchKernelGGL(k_gather_fixup<FMT_FP16>, dim3(g.rows / kGatherRows), dim3(256), 0, s, g, fa);
  else
    hipLaunchKernelGGL(k_gather_fixup<FMT_BF16_SPLIT>, dim3(g.rows / kGatherRows), dim3(256), 0, s, g, fa);
  return hipGetLastError();
}

// ---- launch ----------------------------------------------------------------------------------------
// The gathers of one chunk write disjoint column ranges of X and read nothing another gather writes, so only the
// first one keeps the stream's order (it waits for the point sort); the others are dispatched without the queue
// barrier (list_common.h, LIST_LAUNCH / any_order()): six drains of the chip less per chunk, 0.91 -> 0.86 ms for the
// seven launches of the metric's shape.  `order` = 0 keeps the plain in-order launch (taken for all seven when the
// caller asks for a stage event between two gathers: per-kernel timing).
static bool vox_level_is_near(const ListVoxLevel& lv) {
  const int big = lv.W > lv.H ? (lv.W > lv.D ? lv.W : lv.D) : (lv.H > lv.D ? lv.H : lv.D);
  return kDisp * 0.5f * (float)(big - 1) < 0.99f && lv.C >= 16;           // stencil stays within one cell
}

// ride: the scalar level / xyz / padding that this level's kernel writes too (k_gather_vox only), or null
template <int C, int FMT, int F16>
static hipError_t launch_vox_level_t(const GatherParams& g, const ListVoxLevel& lv, int col_off,
                                     hipStream_t s, int order, const TailArgs* ride) {
  using G = VoxGeom<C, MapT<F16>::V>;
  const bool near = vox_level_is_near(lv);
  if (near && gather_box_eligible(g, lv, col_off)) return launch_gather_vox_box(g, lv, col_off, s, order);
  TailArgs none;
  memset(&none, 0, sizeof(none));
  if (near)
    LIST_LAUNCH((k_gather_vox_near<C, FMT, F16>), dim3(g.rows / G::RB), dim3(256), 0, s, order, g, lv, col_off);
  else if (ride) {
    if constexpr (C <= 64) LIST_LAUNCH((k_gather_vox<C, FMT, F16, 1>), dim3(g.rows / G::RB), dim3(256), 0, s, order, g, lv, col_off, *ride);
    else return hipErrorInvalidValue;
  } else
    LIST_LAUNCH((k_gather_vox<C, FMT, F16, 0>), dim3(g.rows / G::RB), dim3(256), 0, s, order, g, lv, col_off, none);
  return hipGetLastError();
}

template <int C, int FMT>
static hipError_t launch_vox_level(const GatherParams& g, const ListVoxLevel& lv, int col_off,
                                   hipStream_t s, int order, const TailArgs* ride) {
  if (lv.dtype == LIST_MAP_F16) {
    if constexpr (C >= 8) return launch_vox_level_t<C, FMT, 1>(g, lv, col_off, s, order, ride);
    else return hipErrorInvalidValue;
  }
  return launch_vox_level_t<C, FMT, 0>(g, lv, col_off, s, order, ride);
}

#ifndef LIST_GATHER_SEQ
#define LIST_GATHER_SEQ 123I45T
#endif
#define LIST_STR2(x) #x
#define LIST_STR(x) LIST_STR2(x)

template <int FMT>
static hipError_t launch_gather_fmt(const GatherParams& g, const FeatLayout& L, const ListQueryArgs& a,
                                    int* nan_tiles, hipStream_t s, bool skip_img) {
  hipError_t e = hipSuccess;
  auto mark = [&](int stage) {
    if (a.stage_events && a.stage_events[stage]) (void)hipEventRecord((hipEvent_t)a.stage_events[stage], s);
  };
  bool timed_apart = false;                             // an event between two gathers: they run one by one
  for (int st = LIST_STAGE_VOX0; st <= LIST_STAGE_IMG; ++st) timed_apart |= a.stage_events && a.stage_events[st];
  const int side = timed_apart ? 0 : any_order();       // launches after the first one
  int order = 0;                                        // the first gather stays in stream order
  TailLevels tl;
  tl.n = 0;
  for (int l = 0; l < LIST_N_VOX_LEVELS; ++l)
    if (a.vox[l].C == 1) { tl.lv[tl.n] = a.vox[l]; tl.off[tl.n] = L.vox_off[l]; ++tl.n; }

  // The scalar level rides along with the vector level of the same grid (gather_pair, TailRide): one scalar level,
  // fp32 in place, and a level of 8 .. 64 channels with its dimensions that takes k_gather_vox.  LIST_TAIL_RIDE=0 keeps
  // k_gather_tail (A/B runs).
  static const bool ride_on = [] { const char* e = getenv("LIST_TAIL_RIDE"); return !(e && e[0] == '0' && e[1] == 0); }();
  int ride_level = -1;
  TailArgs ride;
  memset(&ride, 0, sizeof(ride));
  if (ride_on && tl.n == 1 && tl.lv[0].dtype == LIST_MAP_F32) {
    for (int l = 0; l < LIST_N_VOX_LEVELS && ride_level < 0; ++l) {
      const ListVoxLevel& lv = a.vox[l];
      if (lv.C >= 8 && lv.C <= 64 && lv.D == tl.lv[0].D && lv.H == tl.lv[0].H && lv.W == tl.lv[0].W && !vox_level_is_near(lv) &&
          (int64_t)lv.D * lv.H * lv.W * lv.C < ((int64_t)1 << 31))
        ride_level = l;
    }
    if (ride_level >= 0) {
      ride.l0 = (const float*)tl.lv[0].data; ride.l0_stride = tl.lv[0].image_stride; ride.l0_off = tl.off[0];
      ride.xyz_off = L.xyz_off; ride.F = L.F; ride.nan_tiles = nan_tiles;
    }
  }
  auto vox_level = [&](int l) -> hipError_t {
    const ListVoxLevel& lv = a.vox[l];
    const TailArgs* rd = l == ride_level ? &ride : nullptr;
    switch (lv.C) {
      case 4: return launch_vox_level<4, FMT>(g, lv, L.vox_off[l], s, order, rd);
      case 8: return launch_vox_level<8, FMT>(g, lv, L.vox_off[l], s, order, rd);
      case 16: return launch_vox_level<16, FMT>(g, lv, L.vox_off[l], s, order, rd);
      case 32: return launch_vox_level<32, FMT>(g, lv, L.vox_off[l], s, order, rd);
      case 64: return launch_vox_level<64, FMT>(g, lv, L.vox_off[l], s, order, rd);
      case 128: return launch_vox_level<128, FMT>(g, lv, L.vox_off[l], s, order, rd);
      case 256: return launch_vox_level<256, FMT>(g, lv, L.vox_off[l], s, order, rd);
      default: return hipErrorInvalidValue;
    }
  };
  auto img = [&]() -> hipError_t {
    if (skip_img) return hipSuccess;      // the perceptual block is produced inside fc_0 (fused_fc0_kernels.hip)
    if (a.percep_proj) {          // projected perceptual map: H1 channels, fp16 (fp16 operands) or fp32
      if (FMT == FMT_FP16)
        LIST_LAUNCH((k_gather_img<FMT, 1, 1>), dim3(g.rows / kGatherRows), dim3(256), 0, s, order, g, a.percep_proj,
                    a.trans_mat, a.map_size, a.H1, a.clamp_hi, 0, 0);
      else
        LIST_LAUNCH((k_gather_img<FMT, 0, 1>), dim3(g.rows / kGatherRows), dim3(256), 0, s, order, g, a.percep_proj,
                    a.trans_mat, a.map_size, a.H1, a.clamp_hi, 0, 0);
    } else if (a.img_proj) {      // list_prep_img_proj: img_kept_C sampled channels | H1 projected ones per pixel
      if (FMT == FMT_FP16)
        LIST_LAUNCH((k_gather_img<FMT, 1, 1>), dim3(g.rows / kGatherRows), dim3(256), 0, s, order, g, a.img_map,
                    a.trans_mat, a.map_size, a.img_kept_C + a.H1, a.clamp_hi, L.img_off, a.img_kept_C);
      else
        LIST_LAUNCH((k_gather_img<FMT, 0, 1>), dim3(g.rows / kGatherRows), dim3(256), 0, s, order, g, a.img_map,
                    a.trans_mat, a.map_size, a.img_kept_C + a.H1, a.clamp_hi, L.img_off, a.img_kept_C);
    } else if (a.percep_feat) {
      LIST_LAUNCH(k_copy_percep<FMT>, dim3((g.rows + 255) / 256), dim3(256), 0, s, order, g,
                  a.percep_feat, a.pf_sb, a.pf_sc, a.pf_sn, L.img_C, L.img_off);
    } else if (a.img_dtype == LIST_MAP_F16) {
      LIST_LAUNCH((k_gather_img<FMT, 1>), dim3(g.rows / kGatherRows), dim3(256), 0, s, order, g, a.img_map,
                  a.trans_mat, a.map_size, L.img_C, a.clamp_hi, L.img_off, 0);
    } else {
      LIST_LAUNCH((k_gather_img<FMT, 0>), dim3(g.rows / kGatherRows), dim3(256), 0, s, order, g, a.img_map,
                  a.trans_mat, a.map_size, L.img_C, a.clamp_hi, L.img_off, 0);
    }
    return hipGetLastError();
  };
  auto tail = [&]() -> hipError_t {
    if (ride_level >= 0) return hipSuccess;           // written by the riding level's kernel
    LIST_LAUNCH(k_gather_tail<FMT>, dim3((g.rows + 31) / 32), dim3(256), 0, s, order, g, tl, L.xyz_off, L.F, nan_tiles);
    return hipGetLastError();
  };

  if (timed_apart) {            // level order, one stage event after each (include/list_hip.h, ListStage)
    int vec_level = 0;
    for (int l = 0; l < LIST_N_VOX_LEVELS; ++l) {
      if (a.vox[l].C == 1) continue;
      if ((e = vox_level(l)) != hipSuccess) return e;
      if (vec_level < 5) mark(LIST_STAGE_VOX0 + vec_level);
      ++vec_level;
    }
    for (; vec_level < 5; ++vec_level) mark(LIST_STAGE_VOX0 + vec_level);
    if ((e = img()) != hipSuccess) return e;
    mark(LIST_STAGE_IMG);
    return tail();
  }
  // back to back without barriers: neighbours in the launch sequence share the chip at the boundary (digit = voxel
  // level, I = 2-D gather, T = tail; whatever the sequence does not name follows in level order).  Five sequences
  // measured within 1 % of each other in round 2; the coarse levels first, then the 2-D gather, then the HBM-bound
  // fine levels was the best.  Round 3 (coarse levels on the matrix cores, gather_box_kernels.hip), eleven sequences on
  // one device (tools/ab_gather.sh): with the coarse levels LAST the group takes 0.79-0.83 instead of 0.85-0.87 ms and
  // fc_0 behind it 0.50-0.53 instead of 0.48 ms -- group + fc_0 = 1.34-1.35 ms whatever the order (the X lines the
  // gathers leave dirty in the L2s drain into whatever runs next), the step 2.04-2.10 ms.  Kept: the round-2 order
  // Round 4b: with the 2-D sample inside fc_0 and fc_0 short of its projected K-tiles (fp16 inference forwards) the
  // balance moved: the fine levels FIRST and the matrix-core levels last, 123I45T, takes the fp16 step from 1.944 / 1.951
  // to 1.930 / 1.922 ms (group -0.05, fc_0 +0.02; two interleaved repetitions, tools/r4b_seq_ab.sh); bf16x3 and the training
  // step inside their noise, plain bf16 +0.02.  Now the default.
  // (Round 4, measured and dropped: the seven gathers spread over two to four QUEUES -- side streams forked and joined
  // around this group -- run side by side (group 0.85 -> 0.79 ms) but every launch around them pays for the fork / join
  // (sort, fc_0, tail, the preps of the next step): step 2.08 -> 2.08 ... 2.18 ms over five assignments.)
  bool done[LIST_N_VOX_LEVELS + 2] = {false};
  for (const char* c = LIST_STR(LIST_GATHER_SEQ); ; ++c) {
    const bool rest = *c == 0;
    for (int l = 0; l < LIST_N_VOX_LEVELS + 2; ++l) {
      const bool named = l < LIST_N_VOX_LEVELS ? *c == '0' + l : *c == (l == LIST_N_VOX_LEVELS ? 'I' : 'T');
      if (done[l] || !(rest || named)) continue;
      done[l] = true;
      if (l < LIST_N_VOX_LEVELS) { if (a.vox[l].C == 1) continue; e = vox_level(l); }
      else e = l == LIST_N_VOX_LEVELS ? img() : tail();
      if (e != hipSuccess) return e;
      order = side;
    }
    if (rest) break;
  }
  return hipSuccess;
}

// bit l set: voxel level l goes to the matrix-core gather (the condition launch_vox_level_t dispatches on)
int gather_box_levels(const GatherParams& g, const FeatLayout& L, const ListQueryArgs& a) {
  int mask = 0;
  for (int l = 0; l < LIST_N_VOX_LEVELS; ++l) {
    const ListVoxLevel& lv = a.vox[l];
    const int big = lv.W > lv.H ? (lv.W > lv.D ? lv.W : lv.D) : (lv.H > lv.D ? lv.H : lv.D);
    const bool near = kDisp * 0.5f * (float)(big - 1) < 0.99f && lv.C >= 16;
    if (near && gather_box_eligible(g, lv, L.vox_off[l])) mask |= 1 << l;
  }
  return mask;
}

hipError_t launch_gather(const GatherParams& g, const FeatLayout& L, const ListQueryArgs& a,
                         int* nan_tiles, hipStream_t s, bool skip_img) {
  return g.fmt == FMT_FP16 ? launch_gather_fmt<FMT_FP16>(g, L, a, nan_tiles, s, skip_img)
                           : launch_gather_fmt<FMT_BF16_SPLIT>(g, L, a, nan_tiles, s, skip_img);
}

// ---- diagnostics: X (gather order, hi+lo) -> out[B][F][N] in the reference order ---------------------
__global__ __launch_bounds__(256) void k_features_out(GatherParams g, FeatLayout L,
                                                      float* __restrict__ out, int* __restrict__ nan_tiles) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)g.n_valid * L.Kp;
  if (i >= total) return;
  const int row = (int)(i / L.Kp), kp = (int)(i - (int64_t)row * L.Kp);
  const int kr = ref_index_of(L, kp);
  if (kr < 0) return;
  const int64_t gp = g.p_begin + (g.order ? g.order[row] : row);
  const int b = (int)(gp / g.N);
  const int n = (int)(gp - (int64_t)b * g.N);
  const float v = g.fmt == FMT_FP16 ? h2f(g.x_hi[i]) : bf2f(g.x_hi[xi_off(i)]) + bf2f(g.x_hi[xi_off(i) + kXiLo]);
  out[((int64_t)b * L.F + kr) * g.N + n] = v;
  if (nan_tiles && v != v) nan_tiles[row / kRowTile] = 1;        // same trigger as fc_0's epilogue
}

hipError_t launch_features_out(const GatherParams& g, const FeatLayout& L, float* out, int* nan_tiles,
                               hipStream_t s) {
  const int64_t total = (int64_t)g.n_valid * L.Kp;
  hipLaunchKernelGGL(k_features_out, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, g, L,
                     out, nan_tiles);
  return hipGetLastError();
}

// ---- PerceptualPooling.forward alone: out[B][Ct][N] ---------------------------------------------------
// lanes over points (coalesced writes along N), loop over channel groups.
template <int F16>
__global__ __launch_bounds__(256) void k_percep_pool(ListPoolArgs a) {
  using M = MapT<F16>;
  const int64_t gp = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gp >= (int64_t)a.B * a.N) return;
  const int b = (int)(gp / a.N);
  const int n = (int)(gp - (int64_t)b * a.N);
  const float* q = a.pc + (int64_t)b * a.p_sb + (int64_t)n * a.p_sn;
  const Proj pr = project(a.trans_mat + b * 12, q[0], q[a.p_sc], q[2 * a.p_sc], a.map_size,
                          a.img_C, a.clamp_hi);
  const int64_t bo = (int64_t)b * a.map_size * a.map_size * a.img_C;
  float* o = a.out + (int64_t)b * a.img_C * a.N + n;
  for (int c = 0; c < a.img_C; c += M::V) {
    float r[M::V];
    const typename M::Raw v[4] = {M::load(a.img_map, bo + pr.o00 + c), M::load(a.img_map, bo + pr.o01 + c),
                                  M::load(a.img_map, bo + pr.o10 + c), M::load(a.img_map, bo + pr.o11 + c)};
    tap_mul<M>(v[0], pr.w00, r); tap_fma<M>(v[1], pr.w01, r);
    tap_fma<M>(v[2], pr.w10, r); tap_fma<M>(v[3], pr.w11, r);
    if (any_nan<M::V>(r)) reduce_proj_exact<M>(v, pr, r);
#pragma unroll
    for (int k = 0; k < M::V; ++k) o[(int64_t)(c + k) * a.N] = r[k];
  }
}

hipError_t launch_percep_pool(const ListPoolArgs& a, hipStream_t s) {
  const int64_t P = (int64_t)a.B * a.N;
  if (a.img_dtype == LIST_MAP_F16)
    hipLaunchKernelGGL(k_percep_pool<1>, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL(k_percep_pool<0>, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, s, a);
  return hipGetLastError();
}

}  // namespace list
