// Internal definitions shared by the gfx950 kernels of the LIST SDF query path.
// Not part of the C ABI (that is include/list_hip.h).
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdlib.h>

#include "list_hip.h"

namespace list {

// ---- independent launches -----------------------------------------------------------------------------------------
// A stream keeps its kernels in order with the barrier bit of their AQL packets: each launch waits until every earlier
// packet of the queue has drained and its caches are written back (~7 us of idle chip per boundary).  Launches that
// do not depend on one another -- the seven gathers of a row chunk -- are therefore dispatched with
// hipExtAnyOrderLaunch behind the first of their group: no barrier bit, no drain / write-back / invalidate between
// them.  The next in-order launch waits for all of them, as for any earlier packet.
// (Measured: this removes the boundaries, it does not run kernels of one queue side by side -- eight 16-workgroup
// kernels launched this way still take eight times one; concurrency needs a second stream.)
// LIST_LAUNCH_IN_ORDER (compile-time definition, or the environment variable of that name set to anything but "0"
// when the library makes its first query): every launch in order, for A/B runs and as the switch to pull should a
// runtime mishandle the flag.
#ifdef LIST_LAUNCH_IN_ORDER
inline int any_order() { return 0; }
#else
inline int any_order() {
  static const int flag = [] {
    const char* e = getenv("LIST_LAUNCH_IN_ORDER");
    return (e && e[0] && !(e[0] == '0' && e[1] == 0)) ? 0 : (int)hipExtAnyOrderLaunch;
  }();
  return flag;
}
#endif
#define LIST_LAUNCH(kernel, grid, block, lds, s, order, ...) \
  hipExtLaunchKernelGGL(kernel, grid, block, lds, s, nullptr, nullptr, (order), __VA_ARGS__)

// X stores of the gathers.  Split formats (hi / lo halfs, 4 B per feature): non-temporal -- X is written once and next
// read by fc_0, and should not evict the map lines the gathers re-read from L2 (round 1: 2-D gather 0.30 -> 0.25 ms).
// fp16 X (round 3): PLAIN stores -- the pieces a gather writes (32 B ... 256 B per sample) merge in L2 into whole lines
// before they leave; all gathers plain against all non-temporal on one device, four interleaved runs: gather group
// 0.866 -> 0.841 ms, step 2.110 -> 2.085 ms (no difference in bf16x3, whose pieces are twice as long).
// -DLIST_X_NT_STORES / -DLIST_X_PLAIN_STORES force one policy for A/B runs.
template <bool FP16, typename V>
__device__ __forceinline__ void x_store(const V& v, V* p) {
#if defined(LIST_X_PLAIN_STORES)
  *p = v;
#elif defined(LIST_X_NT_STORES)
  __builtin_nontemporal_store(v, p);
#else
  if (FP16) *p = v;
  else __builtin_nontemporal_store(v, p);
#endif
}

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int kRowTile = 256;       // GEMM BM: workspace rows are padded to this
constexpr int kKTile = 64;          // GEMM BK (largest): feature columns are padded to this
constexpr int kGatherRows = 64;     // query points per gather workgroup
constexpr float kDisp = 0.0722f;    // stencil displacement, network/modules.py:205

// ---- bf16 hi/lo split -------------------------------------------------------------------
// hi = bf16_rne(x), lo = bf16_rne(x - hi): x ~ hi + lo to 16 significant bits.
__device__ __forceinline__ unsigned short f2bf(float x) {
  return __builtin_bit_cast(unsigned short, (__bf16)x);     // v_cvt_pk_bf16_f32, NaN-preserving
}
__device__ __forceinline__ float bf2f(unsigned short h) {
  return __builtin_bit_cast(float, (unsigned)h << 16);
}
// lo plane of x given its hi plane h: bf16(x - h); an infinity has lo = 0 (inf - inf would make hi + lo a NaN)
__device__ __forceinline__ unsigned short bf_lo(float x, unsigned short h) {
  const float d = x - bf2f(h);
  return f2bf(__builtin_isinf(x) ? 0.f : d);
}
__device__ __forceinline__ void split4(const float4& v, uint2& hi, uint2& lo) {
  const unsigned short h0 = f2bf(v.x), h1 = f2bf(v.y), h2 = f2bf(v.z), h3 = f2bf(v.w);
  const unsigned short l0 = bf_lo(v.x, h0), l1 = bf_lo(v.y, h1);
  const unsigned short l2 = bf_lo(v.z, h2), l3 = bf_lo(v.w, h3);
  hi = make_uint2((unsigned)h0 | ((unsigned)h1 << 16), (unsigned)h2 | ((unsigned)h3 << 16));
  lo = make_uint2((unsigned)l0 | ((unsigned)l1 << 16), (unsigned)l2 | ((unsigned)l3 << 16));
}

// ---- fp16 (single plane) -------------------------------------------------------------------
// fp32 -> fp16, round-to-nearest-even, magnitudes above 65504 (and +-inf) saturate, NaN stays NaN
// (v_med3_f32 alone would turn a NaN into -65504: the reference propagates NaNs, so do the stores).
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
__device__ __forceinline__ float sat_h(float x) {
#ifdef LIST_SAT_H_MINMAX          // A/B only: the round-1 clamp (a NaN becomes -65504)
  return fminf(fmaxf(x, -65504.f), 65504.f);
#endif
  const float c = __builtin_amdgcn_fmed3f(x, -65504.f, 65504.f);
  return (x != x) ? x : c;
}
__device__ __forceinline__ unsigned short f2h(float x) {
  return __builtin_bit_cast(unsigned short, (_Float16)sat_h(x));   // v_cvt_f16_f32, RNE
}
__device__ __forceinline__ unsigned f2h2(float a, float b) {       // v_cvt_pk_f16_f32: a -> low half
  const f32x2_t v = {sat_h(a), sat_h(b)};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2_t));
}
__device__ __forceinline__ float h2f(unsigned short h) {
  return (float)__builtin_bit_cast(_Float16, h);
}
__device__ __forceinline__ void put_map(void* out, int64_t i, float v, int f16) {   // scalar fallback paths
  if (f16) ((unsigned short*)out)[i] = f2h(v);
  else ((float*)out)[i] = v;
}
__device__ __forceinline__ uint2 half4(const float4& v) {
#ifdef LIST_HALF4_SCALAR        // A/B: four scalar conversions + shifts instead of two v_cvt_pk_f16_f32
  return make_uint2((unsigned)f2h(v.x) | ((unsigned)f2h(v.y) << 16),
                    (unsigned)f2h(v.z) | ((unsigned)f2h(v.w) << 16));
#else
  return make_uint2(f2h2(v.x, v.y), f2h2(v.z, v.w));
#endif
}
// no clamp: for values known to lie inside the fp16 range (interpolations of fp16 maps: a convex combination
// of halfs is at most 65504 (1 + 2^-22), which still rounds to 65504); NaN stays NaN
__device__ __forceinline__ uint2 half4_inrange(const float4& v) {
  const f32x2_t a = {v.x, v.y}, b = {v.z, v.w};
  return make_uint2(__builtin_bit_cast(unsigned, __builtin_convertvector(a, f16x2_t)),
                    __builtin_bit_cast(unsigned, __builtin_convertvector(b, f16x2_t)));
}
// ReLU as torch computes it: a NaN stays a NaN (v_max_f32 would return 0)
__device__ __forceinline__ float relu_nan(float x) { return x < 0.f ? 0.f : x; }
enum { FMT_BF16_SPLIT = 0, FMT_FP16 = 1 };     // element format of X / H / packed weights

// ---- hi / lo planes of the long-K operands (X and the packed fc_0 weight), bf16 split formats ----------------
// The two planes are INTERLEAVED in 64-byte blocks: element `off` (= row * Kp + column, Kp % 32 == 0) has its hi half
// at xi_off(off) and its lo half 32 elements further, so the 32 hi and 32 lo halfs of a 32-column block are one
// 128-byte line.  fc_0 then stages a K-tile of 32 columns as full lines (separate planes gave 64-B half-line
// LDS-DMA segments, which cap the L2 -> LDS rate) into the same LDS image the single-plane schedule uses.  The
// region is the former [hi plane | lo plane] pair, which must be contiguous.  fp16 (one plane) is not affected.
__device__ __forceinline__ int64_t xi_off(int64_t off) { return ((off >> 5) << 6) | (off & 31); }
constexpr int kXiLo = 32;                      // elements from a hi half to its lo half

// store 4 consecutive features of one row (off % 4 == 0)
template <int FMT>
__device__ __forceinline__ void store_feat4(unsigned short* x_hi, unsigned short* x_lo, int64_t off,
                                            const float4& v) {
  typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
  // (store policy: x_store above)
  if (FMT == FMT_FP16) {
    const uint2 h = half4(v);
    x_store<true>((u32x2){h.x, h.y}, (u32x2*)(x_hi + off));
  } else {
    uint2 hi, lo;
    split4(v, hi, lo);
    (void)x_lo;
    unsigned short* d = x_hi + xi_off(off);
    x_store<false>((u32x2){hi.x, hi.y}, (u32x2*)d);
    x_store<false>((u32x2){lo.x, lo.y}, (u32x2*)(d + kXiLo));
  }
}
template <int FMT>
__device__ __forceinline__ void store_feat1(unsigned short* x_hi, unsigned short* x_lo, int64_t off,
                                            float v) {
  if (FMT == FMT_FP16) {
    x_hi[off] = f2h(v);
  } else {
    const unsigned short h = f2bf(v);
    (void)x_lo;
    x_hi[xi_off(off)] = h;
    x_hi[xi_off(off) + kXiLo] = bf_lo(v, h);
  }
}

// ---- XCD-aware workgroup order ---------------------------------------------------------------
// Workgroups are dealt round-robin over the 8 XCDs (b and b+8 share an XCD and its 4 MB L2).  This
// bijection hands XCD k the k-th CONTIGUOUS eighth of the logical block range, so that -- with rows
// in Morton order -- every L2 serves one compact region of the maps instead of all of them.
// Placement only affects speed, never results.
__device__ __forceinline__ int xcd_contiguous_block(int bid, int nblocks) {
  const int q = nblocks / 8, r = nblocks % 8, xcd = bid % 8;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
}

// ---- feature layout ---------------------------------------------------------------------
// Column order of the gathered feature matrix X[row][Kp] ("gather order"):
//   [perceptual img_C]  [vector voxel levels (C%4==0): level-major, stencil j, channel c]
//   [scalar voxel levels (C==1): stencil j]  [xyz]  [zero pad to a multiple of kKTile]
// The reference order (network/modules.py:270-275) is k = (cbase_L + c)*7 + j | img | xyz;
// fc_0's columns are permuted once (list_prep_mlp_weights) so no shuffle happens per point.
struct FeatLayout {
  int vox_C[LIST_N_VOX_LEVELS];
  int vox_off[LIST_N_VOX_LEVELS];     // first column of the level in gather order
  int vox_cbase[LIST_N_VOX_LEVELS];   // first channel of the level in the reference concat
  int vox_ctotal;
  int img_off, img_C;
  int xyz_off;
  int F;                               // 7*vox_ctotal + img_C + 3
  int Kp;                              // F padded to kKTile
};

// returns false if a channel count is unsupported
inline bool make_layout(const int32_t vox_C[LIST_N_VOX_LEVELS], int32_t img_C, FeatLayout* L) {
  int cbase = 0, col = 0;
  for (int l = 0; l < LIST_N_VOX_LEVELS; ++l) {
    const int C = vox_C[l];
    if (C < 1 || (C != 1 && (C % 4) != 0)) return false;
    L->vox_C[l] = C;
    L->vox_cbase[l] = cbase;
    cbase += C;
  }
  L->vox_ctotal = cbase;
  // the perceptual block comes FIRST: fc_0 can then leave it out by starting its K loop at column img_C (inference
  // with a projected perceptual map, list_prep_percep_proj) without splitting a K-tile
  if (img_C < 0 || (img_C % 4) != 0) return false;
  L->img_off = col; L->img_C = img_C; col += img_C;
  for (int l = 0; l < LIST_N_VOX_LEVELS; ++l)
    if (L->vox_C[l] != 1) { L->vox_off[l] = col; col += LIST_N_STENCIL * L->vox_C[l]; }
  for (int l = 0; l < LIST_N_VOX_LEVELS; ++l)
    if (L->vox_C[l] == 1) { L->vox_off[l] = col; col += LIST_N_STENCIL; }
  L->xyz_off = col; col += 3;
  L->F = col;
  L->Kp = (col + kKTile - 1) / kKTile * kKTile;
  return true;
}

// gather-order column -> reference feature index (or -1 for padding)
__host__ __device__ inline int ref_index_of(const FeatLayout& L, int kp) {
  if (kp >= L.F) return -1;
  if (kp >= L.xyz_off) return LIST_N_STENCIL * L.vox_ctotal + L.img_C + (kp - L.xyz_off);
  if (kp >= L.img_off && kp < L.img_off + L.img_C)
    return LIST_N_STENCIL * L.vox_ctotal + (kp - L.img_off);
  for (int l = 0; l < LIST_N_VOX_LEVELS; ++l) {
    const int C = L.vox_C[l];
    const int r = kp - L.vox_off[l];
    if (r >= 0 && r < LIST_N_STENCIL * C) {
      const int j = r / C, c = r - j * C;
      return (L.vox_cbase[l] + c) * LIST_N_STENCIL + j;
    }
  }
  return -1;
}

// ---- packed MLP parameters (output of list_prep_mlp_weights) --------------------------------
struct PackedMlp {
  size_t w0_hi, w0_lo, w1_hi, w1_lo, w2_hi, w2_lo;   // byte offsets, bf16 [N][K] row-major
  size_t b0, b1, b2, w3, b3;                         // byte offsets, float
  size_t total;
};
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
inline PackedMlp packed_mlp_layout(int Kp, int H1, int H2, int H3) {
  PackedMlp p;
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t r = o; o = align_up(o + bytes, 256); return r; };
  p.w0_hi = take((size_t)H1 * Kp * 2); p.w0_lo = take((size_t)H1 * Kp * 2);
  p.w1_hi = take((size_t)H2 * H1 * 2); p.w1_lo = take((size_t)H2 * H1 * 2);
  p.w2_hi = take((size_t)H3 * H2 * 2); p.w2_lo = take((size_t)H3 * H2 * 2);
  p.b0 = take((size_t)H1 * 4); p.b1 = take((size_t)H2 * 4); p.b2 = take((size_t)H3 * 4);
  p.w3 = take((size_t)H3 * 4); p.b3 = take(4);
  p.total = o;
  return p;
}

// ---- transposed weight copies for the backward (output of list_prep_mlp_weights_bwd) ---------------
// dgrad runs on the same NT kernel: dH[P][K_in] = dZ[P][N_out] . (W^T)[K_in][N_out]^T, so each layer
// needs W^T [K_in][N_out] (row-major, 16-bit planes).  fc_0's K_in = Kp is padded to 256 rows (zeros).
struct PackedMlpBwd {
  size_t w0t_hi, w0t_lo;      // [KpT][H1],  KpT = Kp rounded up to 256
  size_t w1t_hi, w1t_lo;      // [H1][H2]
  size_t w2t_hi, w2t_lo;      // [H2][H3]
  size_t total;
  int KpT;
};
inline PackedMlpBwd packed_mlp_bwd_layout(int Kp, int H1, int H2, int H3) {
  PackedMlpBwd p;
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t r = o; o = align_up(o + bytes, 256); return r; };
  p.KpT = (Kp + 255) / 256 * 256;
  p.w0t_hi = take((size_t)p.KpT * H1 * 2); p.w0t_lo = take((size_t)p.KpT * H1 * 2);
  p.w1t_hi = take((size_t)H1 * H2 * 2); p.w1t_lo = take((size_t)H1 * H2 * 2);
  p.w2t_hi = take((size_t)H2 * H3 * 2); p.w2t_lo = take((size_t)H2 * H3 * 2);
  p.total = o;
  return p;
}

// ---- backward workspace -----------------------------------------------------------------------------------
constexpr int kSortImagesC = 64, kSortPixCellsC = 8192;      // (= kSortImages, kSortPixCells below; needed above them)
#ifndef LIST_VOX_GATHER_MIN_DENSITY
#define LIST_VOX_GATHER_MIN_DENSITY 2.0
#endif
constexpr int kColsumRows = 256;         // rows per partial of the bias-gradient column sums
constexpr int kWgradMaxSplits = 128;

// map-side gather of the perceptual-map gradient: chunks of its heavy groups (bwd_scatter_kernels.hip, kHeavyChunk =
// 1024 candidates).  Every point is a candidate of at most 4 groups, so the chunks of groups above 1024 candidates
// number at most 4 rows / 1024 whole ones plus one partial chunk per heavy group (<= 4 rows / 1024 of those).
inline int img_heavy_max_chunks(int64_t rows) { return (int)(8 * rows / 1024 + 8); }
inline size_t img_heavy_bytes(int64_t rows, int Ct) {
  return 256 + (size_t)kSortImagesC * kSortPixCellsC * 4 + (size_t)img_heavy_max_chunks(rows) * 8 + 256 +
         (size_t)img_heavy_max_chunks(rows) * 4 * Ct * 4;
}

struct BwdWorkspace {
  size_t scale;                              // float[4]: s, 1/s, sum(dsdf), -
  size_t h3_hi, h3_lo;                       // fc_2 activations, re-evaluated (-DLIST_BWD_REEVAL_FC2 builds only)
  size_t dz3_hi, dz3_lo, dz2_hi, dz2_lo, dz1_hi, dz1_lo;
  size_t dx;                                 // [rows][Kp] fp16 (FP16) or fp32
  size_t slab;                               // wgrad partials, fp32
  size_t colsum;                             // bias-gradient partials
  size_t recs;                               // per-point projection records (2-D gradient)
  size_t vs_keys, vs_bins, vs_sums, vs_recs; // voxel-side gather: sample keys, cell counters, scan sums, records
  size_t img_heavy, img_heavy_bytes;         // heavy groups of the perceptual-map gradient's gather (img_heavy_bytes())
  size_t total;
};
inline int wgrad_nominal_splits(int M, int N) {     // enough workgroups to fill 256 CUs twice
  const int tiles = (M / 256) * ((N + 255) / 256);
  int s = 512 / (tiles > 0 ? tiles : 1);
  return s > kWgradMaxSplits ? kWgradMaxSplits : (s < 1 ? 1 : s);
}
inline size_t wgrad_slab_bytes(int Kp, int H1, int H2, int H3) {
  size_t big = (size_t)H1 * Kp * wgrad_nominal_splits(H1, Kp);
  const size_t b1 = (size_t)H2 * H1 * wgrad_nominal_splits(H2, H1);
  const size_t b2 = (size_t)H3 * H2 * wgrad_nominal_splits(H3, H2);
  if (b1 > big) big = b1;
  if (b2 > big) big = b2;
  return big * 4;
}
inline BwdWorkspace bwd_workspace_layout(int64_t rows, int Kp, int H1, int H2, int H3, bool fp16) {
  BwdWorkspace w;
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t r = o; o = align_up(o + bytes, 256); return r; };
  w.scale = take(16);
#ifdef LIST_BWD_REEVAL_FC2      // (A/B builds only: the backward reads H3 where the forward left it, list_capi.hip)
  w.h3_hi = take((size_t)rows * H3 * 2); w.h3_lo = take((size_t)rows * H3 * 2);
#else
  w.h3_hi = w.h3_lo = 0;
#endif
  w.dz3_hi = take((size_t)rows * H3 * 2); w.dz3_lo = take((size_t)rows * H3 * 2);
  w.dz2_hi = take((size_t)rows * H2 * 2); w.dz2_lo = take((size_t)rows * H2 * 2);
  w.dz1_hi = take((size_t)rows * H1 * 2); w.dz1_lo = take((size_t)rows * H1 * 2);
  w.dx = take((size_t)rows * Kp * (fp16 ? 2 : 4));
  w.slab = take(wgrad_slab_bytes(Kp, H1, H2, H3));
  const int hmax = H1 > H2 ? (H1 > H3 ? H1 : H3) : (H2 > H3 ? H2 : H3);
  w.colsum = take((size_t)((rows + kColsumRows - 1) / kColsumRows) * hmax * 4);
  w.recs = take((size_t)rows * 32);
  w.vs_keys = take((size_t)rows * 8 * 4);
  w.vs_bins = take((size_t)4194304 * 4);
  w.vs_sums = take(1024 * 4);
  w.vs_recs = take((size_t)rows * LIST_N_STENCIL * 16);
  w.img_heavy_bytes = img_heavy_bytes(rows, Kp);          // (the perceptual block is at most Kp columns wide)
  w.img_heavy = take(w.img_heavy_bytes);
  w.total = o;
  return w;
}

// ---- per-chunk workspace --------------------------------------------------------------------
constexpr int kSortCellsPerAxis = 16;                        // Morton cells per axis per image
constexpr int kSortCells = kSortCellsPerAxis * kSortCellsPerAxis * kSortCellsPerAxis;   // 4096
constexpr int kSortImages = kSortImagesC;                    // image slots in the key (b % 64)
constexpr int kSortPixCells = kSortPixCellsC;                // pixel-order bins per image (>= ms * ceil(ms/4))
constexpr int kSortBins = (kSortCells + kSortPixCells) * kSortImages;   // Morton + pixel counters (3 MB)

constexpr int kH3 = 256;                             // fc_2's width: the C API accepts no other (list_capi.hip)
struct Workspace {
  size_t x_hi, x_lo, h1_hi, h1_lo, h2_hi, h2_lo;     // byte offsets
  size_t h3_hi, h3_lo;                               // fc_2 activations of a forward that keeps them (round 3: the
                                                     //   backward's head read a re-evaluation of fc_2 before)
  size_t order, keys, bins;                          // point sort: int32 [rows], [rows], [kSortBins]
  size_t order_img, row_of, keys2;                   // pixel order for the 2-D gather; point -> X row
  size_t nan_tiles;                                  // int32 [rows / 256]: tiles whose fc_0 output holds a NaN
  size_t total;
};
inline size_t workspace_row_bytes(int Kp, int H1, int H2) {
  return (size_t)2 * 2 * ((size_t)Kp + H1 + H2 + kH3) + 21;
}
inline size_t workspace_fixed_bytes() { return (size_t)kSortBins * 4 + 17 * 256; }
inline Workspace workspace_layout(int64_t rows, int Kp, int H1, int H2) {
  Workspace w;
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t r = o; o = align_up(o + bytes, 256); return r; };
  w.x_hi = take((size_t)rows * Kp * 2); w.x_lo = take((size_t)rows * Kp * 2);
  w.h1_hi = take((size_t)rows * H1 * 2); w.h1_lo = take((size_t)rows * H1 * 2);
  w.h2_hi = take((size_t)rows * H2 * 2); w.h2_lo = take((size_t)rows * H2 * 2);
  w.h3_hi = take((size_t)rows * kH3 * 2); w.h3_lo = take((size_t)rows * kH3 * 2);
  w.order = take((size_t)rows * 4); w.keys = take((size_t)rows * 4);
  w.order_img = take((size_t)rows * 4); w.row_of = take((size_t)rows * 4);
  w.keys2 = take((size_t)rows * 4);
  w.bins = take((size_t)kSortBins * 4);
  w.nan_tiles = take((size_t)(rows / kRowTile + 1) * 4);
  w.total = o;
  return w;
}

// ---- kernel parameter blocks ----------------------------------------------------------------
struct GatherParams {
  const float* query; int64_t q_sb, q_sn, q_sc;
  int perm0, perm1, perm2; float scale;
  int N;                      // points per image
  int64_t p_begin;            // first global point of this chunk
  int n_valid;                // valid rows in this chunk
  int rows;                   // padded rows (multiple of kRowTile)
  unsigned short* x_hi; unsigned short* x_lo;
  int Kp;
  int fmt;                    // FMT_BF16_SPLIT or FMT_FP16
  const int* order;           // row -> chunk-local point index (Morton order), or nullptr
  const int* order_img;       // 2-D gather: slot -> chunk-local point index (pixel order), or nullptr
  const int* row_of;          // chunk-local point index -> X row (inverse of `order`)
  float* rowvec; int64_t rv_stride;   // projected perceptual sample: fp32 [rows][H1] row vectors (stride in floats), or null
};

// one product of a grouped launch (k_gemm_nt_pp, EPI_DX): its own operands, K and output, m_tiles row tiles of 256
struct GemmGroup { const char* a; const char* w; void* out; int K, a_rows, m_tiles, lda; };
constexpr int kGemmMaxGroups = LIST_N_IMG_LEVELS;     // (the projections of list_prep_img_proj: one per encoder level)

struct GemmParams {
  const char* a_hi; const char* a_lo;     // [M][K] bf16
  const char* w_hi; const char* w_lo;     // [N][K] bf16
  const float* bias;                      // [N]
  int M, N, K;
  unsigned short* out_hi; unsigned short* out_lo; int ldo;   // EPI_RELU_SPLIT
  float* out_f32; int relu;                                  // EPI_F32
  const float* w3; const float* b3; float* sdf; int n_valid; // EPI_RELU_DOT
  const int* order;                                          // sdf[order[row]] if not null
  int fmt;                                                   // FMT_BF16_SPLIT or FMT_FP16
  const unsigned short* mask; int ldmask;                    // EPI_MASK_SPLIT: keep acc where mask plane != 0
  void* dx; int dx_f16; int n_store;                         // EPI_DX: [M][ldo] fp16/fp32, columns < n_store
  int plain_loop;                                            // diagnostics: never take the ping-pong schedule
  int* nan_tiles;                                            // EPI_RELU_SPLIT: nan_tiles[m0 / 256] = 1 if the tile's output holds a NaN
  const int* tile_gate;                                      // run only the row tiles with tile_gate[m0 / 256] != 0
  int x3i;                                                   // split formats: a_hi / w_hi hold hi and lo interleaved (xi_off); a_lo / w_lo unused
  // ping-pong kernel only (fc_0 without its perceptual block; the projection of the perceptual map):
  int lda, ldw;                                              // row strides of A / W in elements (0: K)
  int a_rows;                                                // rows of A that exist (0: M); staging re-reads the last one beyond
  const char* rowvec; int64_t rowvec_stride;                 // EPI_RELU_SPLIT: fp32 [M][N] row vectors added before the ReLU (byte stride), or null
  // EPI_DX on the ping-pong kernel only: n_groups > 0 = several independent products in ONE launch (the projections of
  // list_prep_img_proj: three small GEMMs, 17 - 19 us each as launches of their own).  M = 256 * sum of m_tiles; N, ldw,
  // ldo, n_store, dx_f16, fmt, x3i are shared; a_hi / w_hi / dx / K / a_rows / lda come from the row tile's group
  GemmGroup grp[kGemmMaxGroups]; int n_groups;
  int k_gap_at, k_gap;                                       // K-tiles (128 operand bytes per row) k_gap_at .. are read k_gap tiles further on in
                                                             //   A and W: fc_0 without the PROJECTED levels of its perceptual block
                                                             //   (list_prep_img_proj); K counts the tiles that are read.  0 / 0: none
};

// EPI_MASK_SPLIT: out = acc where the saved activation is positive (ReLU backward), no bias;
// EPI_DX: plain store of acc (fp16 or fp32) with a column guard (N is padded to the tile).
enum { EPI_RELU_SPLIT = 0, EPI_F32 = 1, EPI_RELU_DOT = 2, EPI_MASK_SPLIT = 3, EPI_DX = 4 };

// out[m][n] = sum_p A[p][m] * B[p][n]  (weight gradients: A = dZ, B = X / H); split over p into slabs
struct GemmTnParams {
  const char* a_hi; const char* a_lo; int lda;     // [P][lda] 16-bit planes
  const char* b_hi; const char* b_lo; int ldb;     // [P][ldb]
  int M, N, P;                                     // M % 256 == 0, N % 8 == 0, P % 256 == 0
  int splits, steps_per_split;                     // K-steps (of BK rows) per split
  float* slab; int ldn;                            // [splits][M][ldn] fp32
  int fmt;
  int b_x3i;                                       // split formats: b_hi holds hi and lo interleaved (xi_off: the X operand); b_lo unused
};

// ---- launchers (defined in the .hip files) ----------------------------------------------------
// n_levels: the first n_levels maps are resized into channels [0, sum C) of a map with Ct channels per pixel
hipError_t launch_prep_img(const ListMap2D maps[LIST_N_IMG_LEVELS], int B, int map_size, int Ct,
                           int f16, void* out, hipStream_t s, int n_levels = LIST_N_IMG_LEVELS);
// list_prep_img_proj (prep_kernels.hip): encoder level [B,C,H,W] fp32 (any strides) -> rows [B*H*W][C], fp16 or fp32
// (n levels in ONE launch: outs[i] receives maps[i])
hipError_t launch_img_level_rows(const ListMap2D* maps, void* const* outs, int n, int B, int f16, hipStream_t s);
// out[b][y][x][coff + n] = sum_l resize(P_l)[b][y][x][n]: P_l = channels-last [B][H_l][W_l][H1] (n_src of them), out =
// map with Ct channels per pixel; f16: out is halfs; src_f16: the P_l are halfs (strides of `src` then count halfs);
// H1 % 64 == 0
hipError_t launch_proj_resize_sum(const ListMap2D* src, int n_src, int B, int map_size, int Ct, int coff, int f16,
                                  void* out, hipStream_t s, int src_f16);
hipError_t launch_transpose_vox(const ListMap3D& m, int B, int f16, void* out, hipStream_t s);
// the levels for which transpose_tile_eligible() holds, all in one launch
bool transpose_tile_eligible(const ListMap3D& m, const void* out);
hipError_t launch_transpose_vox_fused(const ListMap3D* maps, void* const* outs, const int* f16, int n, int B,
                                      hipStream_t s);
hipError_t launch_prep_weights(const ListMlpWeights& w, const FeatLayout& L, const PackedMlp& P,
                               char* packed, hipStream_t s);
hipError_t launch_split_xi(const float* x, unsigned short* out, int64_t n, hipStream_t s);
hipError_t launch_split(const float* x, unsigned short* hi, unsigned short* lo, int64_t n, int fmt,
                        hipStream_t s, int order = 0);
struct SortBuffers { int* order; int* order_img; int* row_of; int* keys; int* keys2; int* bins; };
hipError_t launch_sort_points(const GatherParams& g, const ListQueryArgs& a, const SortBuffers& sb,
                              hipStream_t s);
// nan_tiles: int32 [rows / 256], cleared by the last gather kernel (the flags of fc_0's NaN probe)
// skip_img: the 2-D gather is left out (its columns of X are produced inside the fused fc_0, fused_fc0_kernels.hip)
hipError_t launch_gather(const GatherParams& g, const FeatLayout& L, const ListQueryArgs& a,
                         int* nan_tiles, hipStream_t s, bool skip_img = false);
hipError_t launch_features_out(const GatherParams& g, const FeatLayout& L, float* out, int* nan_tiles,
                               hipStream_t s);
// coarse levels on the matrix cores (gather_box_kernels.hip); eligible: near level, C = 128, fp16 maps and fp16 X
bool gather_box_eligible(const GatherParams& g, const ListVoxLevel& lv, int col_off);
hipError_t launch_gather_vox_box(const GatherParams& g, const ListVoxLevel& lv, int col_off, hipStream_t s, int order);
int gather_box_levels(const GatherParams& g, const FeatLayout& L, const ListQueryArgs& a);     // bitmask of such levels
// exact (reference skip semantics) redo of the voxel and 2-D gathers for the 256-row tiles flagged in tile_flags
hipError_t launch_gather_fixup(const GatherParams& g, const FeatLayout& L, const ListQueryArgs& a,
                               const int* tile_flags, hipStream_t s);
hipError_t launch_percep_pool(const ListPoolArgs& a, hipStream_t s);
hipError_t launch_gemm(const GemmParams& p, int terms, int epi, hipStream_t s);
// fc_0 on a 128 x 512 tile with the perceptual block of A produced on chip (fused_fc0_kernels.hip)
struct FusedFc0Params {
  GemmParams gp;                 // a_hi = X [M][K], w_hi = packed W0 [512][K], bias, M (% 128 == 0), N = 512, K (% 64 == 0),
                                 //   out_hi = H1, ldo, nan_tiles
  GatherParams g;                // the points behind the rows (query, order, perm, scale, N, p_begin, n_valid)
  const void* img_map;           // prepared map [B][ms][ms][Ct]: fp16 (fp16 operands) or fp32 (bf16 formats)
  const float* trans_mat;        // [B][4][3]
  int ms, Ct; float clamp_hi;
  int n_produced;                // leading K-tiles produced on chip (Ct / 64, or Ct / 32 in the bf16 formats); 0: every
                                 //   K-tile from X (tile-shape diagnostic)
  int proj, kept;                // proj = 1 (list_prep_img_proj's map, fp16 operands): Ct = kept + N channels per pixel,
                                 //   n_produced = kept / 64, gp.k_gap_at / k_gap leave the projected levels' K-tiles out and
                                 //   the epilogue adds the sample of the N projected channels
};
bool fused_fc0_eligible(const GemmParams& gp, int img_f16, int img_C);
hipError_t launch_fc0_fused(const FusedFc0Params& fp, int terms, hipStream_t s);
// fc_1 + fc_2 + fc_out in one launch (fp16 operands, H2 = H3 = 256, nothing kept for a backward): gemm_kernels.hip
hipError_t launch_mlp_tail(const GemmParams& fc1, const char* w2, const float* b2, const float* w3, const float* b3,
                           float* sdf, const int* order, int n_valid, hipStream_t s);

// backward: MLP (bwd_mlp_kernels.hip)
hipError_t launch_prep_weights_bwd(const ListMlpWeights& w, const FeatLayout& L, const PackedMlpBwd& P,
                                   char* packed, hipStream_t s);
hipError_t launch_gemm_tn(const GemmTnParams& p, int terms, hipStream_t s);
int wgrad_splits(int M, int N, int P, int terms);
// out[m][col(n)] = inv_scale * sum_s slab[s][m][n]; L != nullptr: n is a gather-order column of fc_0,
// written to its reference column (padding dropped); ldo = row stride of out
hipError_t launch_wgrad_reduce(const float* slab, int splits, int M, int N, int ldn, const FeatLayout* L,
                               const float* scale, float* out, int ldo, hipStream_t s);
hipError_t launch_grad_scale(const float* grad_sdf, int64_t n, int fp16, float* scale, float* db3,
                             float* partial, hipStream_t s);
hipError_t launch_head(const float* grad_sdf, const int* order, int n_valid, int rows, int H3,
                       const unsigned short* h3_hi, const float* w3, const float* scale,
                       unsigned short* dz_hi, unsigned short* dz_lo, int fmt, hipStream_t s);
// out[n] = f * sum_r w_r * Z[r][n]; w_r = grad_sdf[order[r]] (rows < n_valid) if weights else 1;
// f = scale[1] if use_inv_scale else 1
hipError_t launch_colsum(const unsigned short* z_hi, const unsigned short* z_lo, int rows, int n_valid,
                         int N, int fmt, const float* grad_sdf, const int* order, const float* scale,
                         int use_inv_scale, float* partial, float* out, hipStream_t s);

// backward: maps (bwd_scatter_kernels.hip)
struct ScatterParams {
  GatherParams g;             // points (x_hi/x_lo unused)
  const void* dx; int dx_f16; // [rows][Kp]
  const float* scale;         // [0] = s, [1] = 1/s
  int forked;                 // the adjoint forms run side by side on auxiliary streams
};
// buffers of the voxel-side gather (bwd_scatter_kernels.hip); bins == nullptr disables it
constexpr int64_t kVoxGatherMaxBins = 4194304;      // cells (B * D * H * W) a level may have
constexpr double kVoxGatherMinDensity = LIST_VOX_GATHER_MIN_DENSITY;   // samples per cell
// mode: ListQueryGradArgs.vox_adjoint; h16 / h16w: scratch for the fp16 image of a direct level / of the two window
// levels (packed-half atomics), or null
struct VoxGatherBuffers { int* keys; int* bins; int* sums; void* recs; int mode; void* h16; size_t h16_bytes; void* h16w; size_t h16w_bytes; };
// the three adjoint forms may run on different streams (gather / direct atomics / LDS windows)
// matrix-core adjoint of an 8^3-class level (bwd_box_kernels.hip); pk_scale: the scale of the level's fp16 image
bool scatter_box_eligible(const ScatterParams& sp, const ListVoxLevel& gv, int col_off, float pk_scale);
// ... and for the formats whose dX is fp32 (bwd_box_split_kernels.hip: bf16 hi + lo operands, fp32 flush)
bool scatter_box_split_eligible(const ScatterParams& sp, const ListVoxLevel& gv, int col_off);
hipError_t launch_scatter_vox_box_split(const ScatterParams& sp, const ListVoxLevel& gv, int col_off, hipStream_t s);
bool scatter_f32_diagnostic();      // LIST_SCATTER_F32=1: the window levels flush fp32 atomics (both kernels; tests)
hipError_t launch_scatter_vox_box(const ScatterParams& sp, const ListVoxLevel& gv, int col_off, _Float16* img16,
                                  hipStream_t s);
#ifdef LIST_BWD_KNOCKOUT    // diagnostic build (wrong gradients): LIST_BWD_SKIP = bit mask of forked-phase stages left out --
// 1 dW0, 2 direct-atomic levels, 4 first window level (16^3), 8 second window level (8^3), 16 voxel-side gather, 32 image
inline int bwd_knockout() { static const int k = [] { const char* e = getenv("LIST_BWD_SKIP"); return e ? atoi(e) : 0; }(); return k; }
#else
constexpr int bwd_knockout() { return 0; }
#endif
struct ScatterStreams { hipStream_t gather, direct, window, window2; };
hipError_t launch_scatter_vox(const ScatterParams& sp, const FeatLayout& L, const ListQueryArgs& a,
                              const ListVoxLevel grad_vox[LIST_N_VOX_LEVELS], const VoxGatherBuffers& vb,
                              const ScatterStreams& st);
// map_f16: grad_img_map receives halfs at the gradient scale (the intermediate of the adjoint resize, fp16 operands)
// heavy / heavy_bytes: scratch of the map-side gather's heavy groups (img_heavy_bytes(); null: every group is walked by
// its own workgroup)
hipError_t launch_img_grad(const ScatterParams& sp, const FeatLayout& L, const ListQueryArgs& a,
                           const int* bins_pix, int nslots, void* recs, float* grad_img_map, int map_f16,
                           float* grad_trans_mat, void* const* stage_events, hipStream_t s, void* heavy = nullptr,
                           size_t heavy_bytes = 0);
hipError_t launch_rows_to_grad(const ScatterParams& sp, int img_off, int C, int B, int* row_of_scratch, float* out,
                               int64_t sb, int64_t sc, int64_t sn, hipStream_t s);
hipError_t launch_grad_to_rows(const float* src, int64_t sb, int64_t sc, int64_t sn, int B, int N, int C, float* dx,
                               float* scale, hipStream_t s);
hipError_t launch_img_grad_to_levels(const float* grad_img_map, int B, int map_size, int Ct,
                                     const ListMap2D grads[LIST_N_IMG_LEVELS], hipStream_t s, int map_f16 = 0,
                                     const float* scale = nullptr);

}  // namespace list
