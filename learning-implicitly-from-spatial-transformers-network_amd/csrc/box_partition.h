// Shared by the matrix-core gather of the coarse voxel levels (gather_box_kernels.hip) and its adjoint
// (bwd_box_kernels.hip): per-axis weight records, the run descriptor, and the segment tree that cuts a workgroup's 64
// Morton-consecutive points into aligned power-of-two runs whose voxel box fits the LDS box.
#pragma once
#include "list_common.h"
#include "point_math.h"

namespace list {

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4v;

struct AxisW { int i0; float w0, w1; };                     // base index; w1 = 0 where the +1 tap is skipped
struct RunBox { int lo, n, b, count; };                     // lo / n: x | y << 8 | z << 16 (n = 0: no valid point)

__device__ __forceinline__ s16x4 tr_read16(const char* l) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)l);
}
__device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }

// fp32 pair -> packed fp16 pair (a in the low half), RNE, no clamp (weights lie in [0, 1])
__device__ __forceinline__ unsigned pk_h2(float a, float b) {
  const f32x2_t v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2_t));
}

// variant of axis `ax` under stencil point j: 0 = centre coordinate, 1 = -d, 2 = +d (network/modules.py:205-214)
__device__ __forceinline__ int variant_of(int ax, int j) { return j == 2 * ax + 1 ? 1 : (j == 2 * ax + 2 ? 2 : 0); }

// ---- partition of the 64 points into runs (one wave, one point per lane) ----------------------------------------
// Tap ranges travel as minima of 16-bit fields (an upper bound hi as 255 - hi): f0 = lo_x | lo_y << 16,
// f1 = lo_z | (255 - hi_x) << 16, f2 = (255 - hi_y) | (255 - hi_z) << 16; images as bmin and ~bmax.  A point that is
// not valid carries the neutral element everywhere.
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_min16(unsigned a, unsigned b) {
  return __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
struct SegBox { unsigned f0, f1, f2; int bmin, nbmax; };
template <int STAGE>
__device__ __forceinline__ int seg_xchg(int v) {
  // the partner half of the aligned 2^(STAGE+1) segment: every lane of a half holds the half's value already, so
  // any lane of the other half will do (quad permutes, then the row mirrors, then across rows)
  if (STAGE == 0) return __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, true);      // quad_perm [1,0,3,2]
  if (STAGE == 1) return __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, true);      // quad_perm [2,3,0,1]
  if (STAGE == 2) return __builtin_amdgcn_update_dpp(v, v, 0x141, 0xf, 0xf, true);     // row_half_mirror
  if (STAGE == 3) return __builtin_amdgcn_update_dpp(v, v, 0x140, 0xf, 0xf, true);     // row_mirror
  return __shfl_xor(v, 1 << STAGE);
}
template <int STAGE>
__device__ __forceinline__ void seg_merge(SegBox& b) {
  b.f0 = pk_min16(b.f0, (unsigned)seg_xchg<STAGE>((int)b.f0));
  b.f1 = pk_min16(b.f1, (unsigned)seg_xchg<STAGE>((int)b.f1));
  b.f2 = pk_min16(b.f2, (unsigned)seg_xchg<STAGE>((int)b.f2));
  b.bmin = min(b.bmin, seg_xchg<STAGE>(b.bmin));
  b.nbmax = min(b.nbmax, seg_xchg<STAGE>(b.nbmax));
}
__device__ __forceinline__ bool seg_fits(const SegBox& b, int maxrows, int maxkeys) {
  if (b.bmin == INT_MAX) return true;                         // no valid point
  if (b.bmin != ~b.nbmax) return false;                       // two images
  const int lox = b.f0 & 0xffff, loy = b.f0 >> 16, loz = b.f1 & 0xffff;
  const int hix = 255 - (int)(b.f1 >> 16), hiy = 255 - (int)(b.f2 & 0xffff), hiz = 255 - (int)(b.f2 >> 16);
  const int nx = hix - lox + 1, ny = hiy - loy + 1, nz = hiz - loz + 1;
  const int nfw = (hix >> 2) - (lox >> 2) + 1;
  return nx * ny * nz <= maxrows && nfw * ny * nz <= maxkeys;
}

}  // namespace list
