// Per-point coordinate arithmetic shared by the forward gathers and the backward scatters: query
// permutation/scale, grid_sample un-normalisation, trilinear taps, the stencil, the projection.
// (Forward and backward must derive bit-identical taps and weights from the same point.)
#pragma once

#include "list_common.h"

namespace list {

// ---- shared per-point helpers -----------------------------------------------------------------
struct Pt { float x, y, z; int b; bool valid; };

__device__ __forceinline__ Pt load_point(const GatherParams& g, int row) {
  Pt p;
  p.valid = row < g.n_valid;
  const int64_t gp = g.p_begin + (p.valid ? (g.order ? g.order[row] : row) : 0);
  p.b = (int)(gp / g.N);
  const int n = (int)(gp - (int64_t)p.b * g.N);
  const float* q = g.query + (int64_t)p.b * g.q_sb + (int64_t)n * g.q_sn;
  p.x = q[(int64_t)g.perm0 * g.q_sc] * g.scale;     // models.py:91-92: query[:, :, [2,1,0]] * 2
  p.y = q[(int64_t)g.perm1 * g.q_sc] * g.scale;
  p.z = q[(int64_t)g.perm2 * g.q_sc] * g.scale;
  return p;
}

// grid_sampler_unnormalize (align_corners) + clip_coordinates + floor, as ATen computes them:
//   v = ((c + 1) / 2) * (size - 1); v = min(size-1, max(v, 0)); i0 = floor(v)
//   w1 = v - i0 ; w0 = (i0 + 1) - v ; the +1 tap is skipped when i0 + 1 == size (its weight is 0)
// A NaN coordinate: ATen's CPU kernel clips with std::min(size-1, std::max(v, 0)), whose comparisons are
// all false for a NaN, so it comes out as size-1 (pinned by tests/golden/hotpath_edge_nan.npz).
struct Axis { int i0; int has1; float w0, w1; };
__device__ __forceinline__ Axis axis_setup(float c, int size) {
  float v = ((c + 1.f) * 0.5f) * (float)(size - 1);
  v = (v != v) ? (float)(size - 1) : fminf((float)(size - 1), fmaxf(v, 0.f));
  const float f = floorf(v);
  Axis a;
  a.i0 = (int)f;
  a.has1 = (a.i0 + 1 < size) ? 1 : 0;
  a.w1 = v - f;
  a.w0 = (f + 1.f) - v;
  return a;
}

// The 8 taps of one trilinear sample: element offsets (relative to the image/channel base) and
// weights in the accumulation order of the ATen CPU kernel: tnw tne tsw tse bnw bne bsw bse.
// `dead`: bit k set = tap k lies beyond the border (index == size) and is SKIPPED by the reference; the fast
// paths read the border voxel instead and multiply by its exactly-zero weight, which is the same number unless
// that voxel holds +-inf / NaN -- the gathers detect a NaN result and re-reduce with reduce_taps_exact.
struct Taps { int o[8]; float w[8]; int dead; };

__device__ __forceinline__ Taps make_taps(float x, float y, float z, int C, int D, int H, int W) {
  const Axis ax = axis_setup(x, W), ay = axis_setup(y, H), az = axis_setup(z, D);
  const int o000 = ((az.i0 * H + ay.i0) * W + ax.i0) * C;
  const int sx = ax.has1 ? C : 0;
  const int sy = ay.has1 ? W * C : 0;
  const int sz = az.has1 ? H * W * C : 0;
  Taps t;
  t.o[0] = o000;           t.o[1] = o000 + sx;
  t.o[2] = o000 + sy;      t.o[3] = o000 + sy + sx;
  t.o[4] = o000 + sz;      t.o[5] = o000 + sz + sx;
  t.o[6] = o000 + sz + sy; t.o[7] = o000 + sz + sy + sx;
  t.w[0] = ax.w0 * ay.w0 * az.w0; t.w[1] = ax.w1 * ay.w0 * az.w0;
  t.w[2] = ax.w0 * ay.w1 * az.w0; t.w[3] = ax.w1 * ay.w1 * az.w0;
  t.w[4] = ax.w0 * ay.w0 * az.w1; t.w[5] = ax.w1 * ay.w0 * az.w1;
  t.w[6] = ax.w0 * ay.w1 * az.w1; t.w[7] = ax.w1 * ay.w1 * az.w1;
  t.dead = (ax.has1 ? 0 : 0xAA) | (ay.has1 ? 0 : 0xCC) | (az.has1 ? 0 : 0xF0);
  return t;
}

// stencil point j of network/modules.py:205-214: centre, then (-d,+d) along x, y, z
template <int J>
__device__ __forceinline__ void stencil_point(const Pt& p, float& x, float& y, float& z) {
  x = p.x + (J == 1 ? -kDisp : J == 2 ? kDisp : 0.f);
  y = p.y + (J == 3 ? -kDisp : J == 4 ? kDisp : 0.f);
  z = p.z + (J == 5 ? -kDisp : J == 6 ? kDisp : 0.f);
}

// ---- 2-D perceptual pooling ---------------------------------------------------------------------
// network/modules.py:37-47 per point.  torch.matmul evaluates the K=4 dot product as an fma chain
// in k order (oracle/list_oracle.py project_points, checked bit-for-bit).
// `dead`: bit 0..3 set = tap 00 / 01 / 10 / 11 is outside the map (zeros padding: the reference adds nothing)
struct Proj { int o00, o01, o10, o11; float w00, w01, w10, w11; int dead; };

__device__ __forceinline__ float clamp_keep_nan(float v, float hi) {
  return (v != v) ? v : fminf(fmaxf(v, 0.f), hi);
}

__device__ __forceinline__ Proj project(const float* __restrict__ T, float px, float py, float pz,
                                        int ms, int Ct, float clamp_hi) {
  float X = px * T[0], Y = px * T[1], Z = px * T[2];
  X = fmaf(py, T[3], X); Y = fmaf(py, T[4], Y); Z = fmaf(py, T[5], Z);
  X = fmaf(pz, T[6], X); Y = fmaf(pz, T[7], Y); Z = fmaf(pz, T[8], Z);
  X = X + T[9]; Y = Y + T[10]; Z = Z + T[11];
  const float den = Z + 1e-8f;
  float u = clamp_keep_nan(__fdiv_rn(X, den), clamp_hi);
  float v = clamp_keep_nan(__fdiv_rn(Y, den), clamp_hi);
  const float half = (float)(ms - 1) * 0.5f;
  const float gx = __fdiv_rn(u - half, half), gy = __fdiv_rn(v - half, half);
  const float ix = (gx + 1.f) * half, iy = (gy + 1.f) * half;     // grid_sample unnormalize
  const float fx = floorf(ix), fy = floorf(iy);
  const float wx1 = ix - fx, wx0 = (fx + 1.f) - ix;
  const float wy1 = iy - fy, wy0 = (fy + 1.f) - iy;
  // zeros padding: a tap at index ms only occurs with weight 0 (ix == ms-1); clamp for safety
  const int x0 = min(max((int)fx, 0), ms - 1), y0 = min(max((int)fy, 0), ms - 1);
  const int x1 = min(x0 + 1, ms - 1), y1 = min(y0 + 1, ms - 1);
  Proj r;
  // in-bounds tests of the reference on the UNCLAMPED indices.  With clamp_hi <= ms-1 (the reference's 136 on
  // its 137^2 map) only the +1 taps at ix == ms-1 / iy == ms-1 fall outside, with weight 0; a smaller map under
  // the hard-coded clamp (modules.py:43 with map_size < 137) puts whole samples outside: zeros padding
  const bool x1_out = !(fx + 1.f <= (float)(ms - 1)), y1_out = !(fy + 1.f <= (float)(ms - 1));
  const bool x0_out = !(fx >= 0.f && fx <= (float)(ms - 1)), y0_out = !(fy >= 0.f && fy <= (float)(ms - 1));
  r.dead = ((x0_out || y0_out) ? 1 : 0) | ((x1_out || y0_out) ? 2 : 0) | ((x0_out || y1_out) ? 4 : 0) |
           ((x1_out || y1_out) ? 8 : 0);
  r.o00 = (y0 * ms + x0) * Ct; r.o01 = (y0 * ms + x1) * Ct;
  r.o10 = (y1 * ms + x0) * Ct; r.o11 = (y1 * ms + x1) * Ct;
  r.w00 = wx0 * wy0; r.w01 = wx1 * wy0; r.w10 = wx0 * wy1; r.w11 = wx1 * wy1;
  // an out-of-map tap contributes 0 * weight in the reference: nothing, unless the weight is NaN (NaN coordinates)
  if ((r.dead & 1) && r.w00 == r.w00) r.w00 = 0.f;
  if ((r.dead & 2) && r.w01 == r.w01) r.w01 = 0.f;
  if ((r.dead & 4) && r.w10 == r.w10) r.w10 = 0.f;
  if ((r.dead & 8) && r.w11 == r.w11) r.w11 = 0.f;
  return r;
}

}  // namespace list
