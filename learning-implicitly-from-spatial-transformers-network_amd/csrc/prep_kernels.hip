// Producer-side hand-off kernels (HBM-bound, no MFMA):
//   * 2-D maps: bilinear resize to map_size^2 (align_corners=True) fused with NCHW -> NHWC and
//     the channel concatenation  (replaces F.interpolate x5, network/modules.py:26-35)
//   * 3-D maps: NCDHW -> NDHWC transpose through LDS
//   * MLP parameters: column permutation + K padding + bf16 hi/lo split
#include "list_common.h"
#include "gather_math.h"

namespace list {

// --------------------------------------------------------------------------------------------
// Resize one level into its channel slice of the concatenated channels-last map.
// grid = (B*ms, ceil(C/32)); block = 256.  LDS tile [ms][33] transposes (c,x) -> (x,c) so the
// writes are 128-B runs along channels.
// ATen's formula (upsample_bilinear2d, align_corners): src = dst*(in-1)/(out-1),
// i0 = trunc(src), l1 = src - i0, l0 = 1 - l1, i1 = i0 + (i0 < in-1).
// --------------------------------------------------------------------------------------------
constexpr int kResizeCg = 32;
constexpr int kResizeMaxMs = 320;

__global__ __launch_bounds__(256) void k_prep_img(ListMap2D m, int ms, int Ct, int coff, int f16,
                                                  void* __restrict__ out) {
  __shared__ float tile[kResizeMaxMs * (kResizeCg + 1)];
  const int b = blockIdx.x / ms;
  const int y = blockIdx.x - b * ms;
  const int c0 = blockIdx.y * kResizeCg;
  const int nc = min(kResizeCg, m.C - c0);

  const float sy = ms > 1 ? (float)(m.H - 1) / (float)(ms - 1) : 0.f;
  const float sx = ms > 1 ? (float)(m.W - 1) / (float)(ms - 1) : 0.f;
  const float fy = sy * (float)y;
  const int y0 = min((int)fy, m.H - 1);
  const int y1 = y0 + (y0 < m.H - 1 ? 1 : 0);
  const float wy1 = fy - (float)y0, wy0 = 1.f - wy1;
  const float* base = m.data + (int64_t)b * m.sb;

  for (int idx = threadIdx.x; idx < nc * ms; idx += 256) {
    const int cl = idx / ms;
    const int x = idx - cl * ms;
    const float fx = sx * (float)x;
    const int x0 = min((int)fx, m.W - 1);
    const int x1 = x0 + (x0 < m.W - 1 ? 1 : 0);
    const float wx1 = fx - (float)x0, wx0 = 1.f - wx1;
    const float* pc = base + (int64_t)(c0 + cl) * m.sc;
    const float v00 = pc[(int64_t)y0 * m.sh + (int64_t)x0 * m.sw];
    const float v01 = pc[(int64_t)y0 * m.sh + (int64_t)x1 * m.sw];
    const float v10 = pc[(int64_t)y1 * m.sh + (int64_t)x0 * m.sw];
    const float v11 = pc[(int64_t)y1 * m.sh + (int64_t)x1 * m.sw];
    const float top = v00 * wx0 + v01 * wx1;
    const float bot = v10 * wx0 + v11 * wx1;
    tile[x * (kResizeCg + 1) + cl] = top * wy0 + bot * wy1;
  }
  __syncthreads();
  const int64_t orow = ((int64_t)(b * ms + y) * ms) * Ct + coff + c0;
  for (int idx = threadIdx.x; idx < ms * kResizeCg; idx += 256) {
    const int x = idx / kResizeCg;
    const int cl = idx - x * kResizeCg;
    if (cl < nc) put_map(out, orow + (int64_t)x * Ct + cl, tile[x * (kResizeCg + 1) + cl], f16);
  }
}

// Fast path (W-contiguous source rows, C % 32 == 0): a workgroup owns a tile of RY output rows x one
// x-range x 32 channels.  It stages the source rows the tile needs in LDS transposed to [row][x][c]
// (global reads coalesced along x), then every thread produces 4 (fp32) or 8 (fp16) channels of one
// output pixel from ds_read_b128s and issues one 16-B store; consecutive lanes = consecutive channel
// groups = contiguous 128-B (64-B) runs.  RY > 1 for the up-sampled levels (14, 28, 56 px -> 137:
// consecutive output rows share their source rows), an x split for the widest level keeps the tile in
// 32 KB of LDS.  Arithmetic and rounding are identical to k_prep_img.
constexpr int kTileG = 32;

template <int F16>
__global__ __launch_bounds__(256) void k_prep_img_tile(ListMap2D m, int ms, int Ct, int coff,
                                                       void* __restrict__ out, int RY, int XS, int WT) {
  extern __shared__ __attribute__((aligned(16))) float rows[];     // [nr][WT][G + 4]
  constexpr int G = kTileG, S = G + 4;
  const int nyb = (ms + RY - 1) / RY, nxo = (ms + XS - 1) / XS;
  int bid = blockIdx.x;
  const int xs = bid % XS; bid /= XS;
  const int yb = bid % nyb;
  const int b = bid / nyb;
  const int c0 = blockIdx.y * G;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

  const float sy = ms > 1 ? (float)(m.H - 1) / (float)(ms - 1) : 0.f;
  const float sx = ms > 1 ? (float)(m.W - 1) / (float)(ms - 1) : 0.f;
  const int y_first = yb * RY, y_last = min(y_first + RY, ms) - 1;
  const int x_first = xs * nxo, x_last = min(x_first + nxo, ms) - 1;
  if (x_first > x_last) return;
  const int ys0 = min((int)(sy * (float)y_first), m.H - 1);
  const int ys1 = min((int)(sy * (float)y_last) + 1, m.H - 1);
  const int xs0 = min((int)(sx * (float)x_first), m.W - 1);
  const int xs1 = min((int)(sx * (float)x_last) + 1, m.W - 1);
  const int nr = ys1 - ys0 + 1, wt = xs1 - xs0 + 1;

  const float* base = m.data + (int64_t)b * m.sb + (int64_t)c0 * m.sc + xs0;
  for (int r = 0; r < nr; ++r) {
    const float* srow = base + (int64_t)(ys0 + r) * m.sh;
    float* drow = rows + r * WT * S;
#pragma unroll
    for (int ci = 0; ci < G / 4; ++ci) {
      const int c = wave + 4 * ci;
      for (int x = lane; x < wt; x += 64) drow[x * S + c] = srow[(int64_t)c * m.sc + x];
    }
  }
  __syncthreads();
  constexpr int VO = F16 ? 8 : 4;          // channels per thread: one 16-B store either way
  constexpr int Q = G / VO;
  const int nx = x_last - x_first + 1, ny = y_last - y_first + 1;
  for (int idx = threadIdx.x; idx < ny * nx * Q; idx += 256) {
    const int q = idx % Q;
    const int t = idx / Q;
    const int xi = t % nx, yi = t / nx;
    const int x = x_first + xi, y = y_first + yi;
    const float fy = sy * (float)y;
    const int y0 = min((int)fy, m.H - 1);
    const int y1 = y0 + (y0 < m.H - 1 ? 1 : 0);
    const float wy1 = fy - (float)y0, wy0 = 1.f - wy1;
    const float fx = sx * (float)x;
    const int x0 = min((int)fx, m.W - 1);
    const int x1 = x0 + (x0 < m.W - 1 ? 1 : 0);
    const float wx1 = fx - (float)x0, wx0 = 1.f - wx1;
    const float* r0 = rows + (y0 - ys0) * WT * S;
    const float* r1 = rows + (y1 - ys0) * WT * S;
    float o[VO];
#pragma unroll
    for (int h = 0; h < VO / 4; ++h) {
      const int c = q * VO + 4 * h;
      const float4 v00 = *(const float4*)(r0 + (x0 - xs0) * S + c);
      const float4 v01 = *(const float4*)(r0 + (x1 - xs0) * S + c);
      const float4 v10 = *(const float4*)(r1 + (x0 - xs0) * S + c);
      const float4 v11 = *(const float4*)(r1 + (x1 - xs0) * S + c);
      o[4 * h + 0] = (v00.x * wx0 + v01.x * wx1) * wy0 + (v10.x * wx0 + v11.x * wx1) * wy1;
      o[4 * h + 1] = (v00.y * wx0 + v01.y * wx1) * wy0 + (v10.y * wx0 + v11.y * wx1) * wy1;
      o[4 * h + 2] = (v00.z * wx0 + v01.z * wx1) * wy0 + (v10.z * wx0 + v11.z * wx1) * wy1;
      o[4 * h + 3] = (v00.w * wx0 + v01.w * wx1) * wy0 + (v10.w * wx0 + v11.w * wx1) * wy1;
    }
    const int64_t oi = ((int64_t)(b * ms + y) * ms + x) * Ct + coff + c0 + q * VO;
    if (F16) {
      const uint2 lo = half4(make_float4(o[0], o[1], o[2], o[3]));
      const uint2 hi = half4(make_float4(o[VO - 4], o[VO - 3], o[VO - 2], o[VO - 1]));
      *(uint4*)((unsigned short*)out + oi) = make_uint4(lo.x, lo.y, hi.x, hi.y);
    } else {
      *(float4*)((float*)out + oi) = make_float4(o[0], o[1], o[2], o[3]);
    }
  }
}

static bool try_prep_img_tile(const ListMap2D& m, int B, int ms, int Ct, int coff, int f16, void* out,
                              hipStream_t s, hipError_t* e) {
  const int align = f16 ? 8 : 4;
  if (m.sw != 1 || (m.C % kTileG) != 0 || (coff % align) != 0 || (Ct % align) != 0 || ms < 2) return false;
  const float sy = (float)(m.H - 1) / (float)(ms - 1), sx = (float)(m.W - 1) / (float)(ms - 1);
  // rows per tile: as many as keep the staged source rows within ~32 KB, at most 8
  int best_ry = 0, best_xs = 1, best_nr = 0, best_wt = 0;
  for (int xs = 1; xs <= 4 && !best_ry; ++xs) {
    const int nxo = (ms + xs - 1) / xs;
    int wt = (int)(sx * (float)nxo) + 3;
    if (wt > m.W) wt = m.W;
    for (int ry = 8; ry >= 1; --ry) {
      int nr = (int)(sy * (float)(ry - 1)) + 3;
      if (nr > m.H) nr = m.H;
      if ((size_t)nr * wt * (kTileG + 4) * sizeof(float) <= 32768) { best_ry = ry; best_xs = xs; best_nr = nr; best_wt = wt; break; }
    }
  }
  if (!best_ry) return false;
  const size_t lds = (size_t)best_nr * best_wt * (kTileG + 4) * sizeof(float);
  const int nyb = (ms + best_ry - 1) / best_ry;
  const dim3 grid((unsigned)(B * nyb * best_xs), m.C / kTileG);
  if (f16)
    hipLaunchKernelGGL(k_prep_img_tile<1>, grid, dim3(256), lds, s, m, ms, Ct, coff, out, best_ry, best_xs, best_wt);
  else
    hipLaunchKernelGGL(k_prep_img_tile<0>, grid, dim3(256), lds, s, m, ms, Ct, coff, out, best_ry, best_xs, best_wt);
  *e = hipGetLastError();
  return true;
}

// Channels-last source (sc == 1, e.g. a ResNet run with memory_format=torch.channels_last): no
// transposition is needed -- every thread resizes 4 (fp32 out) or 8 (fp16 out) consecutive channels
// of one output pixel from four coalesced 16/32-B tap reads.  Same arithmetic as k_prep_img.
template <int F16>
__global__ __launch_bounds__(256) void k_prep_img_nhwc(ListMap2D m, int B, int ms, int Ct, int coff,
                                                       void* __restrict__ out) {
  constexpr int VO = F16 ? 8 : 4;
  const int Q = m.C / VO;
  const int64_t total = (int64_t)B * ms * ms * Q;
  const float sy = ms > 1 ? (float)(m.H - 1) / (float)(ms - 1) : 0.f;
  const float sx = ms > 1 ? (float)(m.W - 1) / (float)(ms - 1) : 0.f;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int q = (int)(idx % Q);
    int64_t t = idx / Q;
    const int x = (int)(t % ms); t /= ms;
    const int y = (int)(t % ms);
    const int b = (int)(t / ms);
    const float fy = sy * (float)y, fx = sx * (float)x;
    const int y0 = min((int)fy, m.H - 1), x0 = min((int)fx, m.W - 1);
    const int y1 = y0 + (y0 < m.H - 1 ? 1 : 0), x1 = x0 + (x0 < m.W - 1 ? 1 : 0);
    const float wy1 = fy - (float)y0, wy0 = 1.f - wy1, wx1 = fx - (float)x0, wx0 = 1.f - wx1;
    const float* base = m.data + (int64_t)b * m.sb + q * VO;
    const float* p00 = base + (int64_t)y0 * m.sh + (int64_t)x0 * m.sw;
    const float* p01 = base + (int64_t)y0 * m.sh + (int64_t)x1 * m.sw;
    const float* p10 = base + (int64_t)y1 * m.sh + (int64_t)x0 * m.sw;
    const float* p11 = base + (int64_t)y1 * m.sh + (int64_t)x1 * m.sw;
    float o[VO];
#pragma unroll
    for (int h = 0; h < VO / 4; ++h) {
      const float4 v00 = *(const float4*)(p00 + 4 * h), v01 = *(const float4*)(p01 + 4 * h);
      const float4 v10 = *(const float4*)(p10 + 4 * h), v11 = *(const float4*)(p11 + 4 * h);
      o[4 * h + 0] = (v00.x * wx0 + v01.x * wx1) * wy0 + (v10.x * wx0 + v11.x * wx1) * wy1;
      o[4 * h + 1] = (v00.y * wx0 + v01.y * wx1) * wy0 + (v10.y * wx0 + v11.y * wx1) * wy1;
      o[4 * h + 2] = (v00.z * wx0 + v01.z * wx1) * wy0 + (v10.z * wx0 + v11.z * wx1) * wy1;
      o[4 * h + 3] = (v00.w * wx0 + v01.w * wx1) * wy0 + (v10.w * wx0 + v11.w * wx1) * wy1;
    }
    const int64_t oi = ((int64_t)(b * ms + y) * ms + x) * Ct + coff + q * VO;
    if (F16) {
      const uint2 lo = half4(make_float4(o[0], o[1], o[2], o[3]));
      const uint2 hi = half4(make_float4(o[VO - 4], o[VO - 3], o[VO - 2], o[VO - 1]));
      *(uint4*)((unsigned short*)out + oi) = make_uint4(lo.x, lo.y, hi.x, hi.y);
    } else {
      *(float4*)((float*)out + oi) = make_float4(o[0], o[1], o[2], o[3]);
    }
  }
}

static bool try_prep_img_nhwc(const ListMap2D& m, int B, int ms, int Ct, int coff, int f16, void* out,
                              hipStream_t s, hipError_t* e) {
  const int vo = f16 ? 8 : 4;
  if (m.sc != 1 || (m.C % vo) != 0 || (coff % vo) != 0 || (Ct % vo) != 0 || (m.sw % 4) != 0 ||
      (m.sh % 4) != 0 || (m.sb % 4) != 0 || (reinterpret_cast<uintptr_t>(m.data) & 15) != 0)
    return false;
  const int64_t total = (int64_t)B * ms * ms * (m.C / vo);
  int64_t blocks = (total + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  if (f16)
    hipLaunchKernelGGL(k_prep_img_nhwc<1>, dim3((unsigned)blocks), dim3(256), 0, s, m, B, ms, Ct, coff, out);
  else
    hipLaunchKernelGGL(k_prep_img_nhwc<0>, dim3((unsigned)blocks), dim3(256), 0, s, m, B, ms, Ct, coff, out);
  *e = hipGetLastError();
  return true;
}

// Row-streaming resize, all levels in ONE launch (C % 64 == 0 levels; any strides).
// The bilinear form of ATen is separable in exactly the order it is evaluated:
//   out = (v00 wx0 + v01 wx1) wy0 + (v10 wx0 + v11 wx1) wy1 = top(y0) wy0 + bot(y1) wy1,
// so a thread that owns (output column x, 8 channels) and walks down RY output rows keeps the two
// horizontally interpolated source rows in registers and only fetches a new one when y0 advances:
// on the up-sampled levels (14/28/56 px -> 137: 87 % of the output bytes) that is one fetch per 10 / 5 /
// 2.5 output rows and 3 flops per output element instead of 7 (the tile kernel above spends more issue
// slots on arithmetic and index math than the 16-B store it feeds can hide).  No LDS: the 8 lanes of a
// pixel read 8-channel groups of the same two taps (L1/L2 hits), and write one full 128-B line (fp16:
// 64 channels; fp32: 256 B) per pixel, 8 pixels per wave-instruction.  Same bits as k_prep_img.
#ifndef LIST_PREP_RY
#define LIST_PREP_RY 16
#endif
#ifndef LIST_PREP_RY_MAX
#define LIST_PREP_RY_MAX 32
#endif
#ifndef LIST_PREP_PX
#define LIST_PREP_PX 16
#endif
#ifndef LIST_PREP_THREADS
#define LIST_PREP_THREADS (LIST_PREP_PX * 8)
#endif
constexpr int kRowsPx = LIST_PREP_PX;                 // output columns per workgroup (x 8 channel octets = 128 threads)
constexpr int kRowsCg = 64;                 // channels per workgroup
struct PrepRowsLevel { ListMap2D m; int coff, wg_begin, cgroups, vec, RY, nyt; };
struct PrepRowsArgs { PrepRowsLevel lv[LIST_N_IMG_LEVELS]; int n_levels, B, ms, Ct, nxt; };

template <int F16>
__global__ __launch_bounds__(LIST_PREP_THREADS) void k_prep_img_rows(PrepRowsArgs a, void* __restrict__ out) {
  // level-major block order: consecutive workgroups are x tiles of one (level, channel group, row block), 32 KB
  // apart in the output.  (Pixel-tile-major -- the 16 channel groups of a pixel tile side by side, so that a
  // pixel's 2 KB leave together -- measured 0.73 ms instead of 0.19: the writes of a moment then fall on few
  // memory channels.)
  // XCD-contiguous: workgroups are dealt round-robin over the 8 XCDs, so consecutive block ids -- neighbouring x
  // tiles, which read the same source lines -- would sit behind 8 different L2s and each fetch them again (PMC:
  // 0.64 GB fetched for 0.15 GB of source).  Every XCD gets one contiguous eighth of the logical order instead.
#ifdef LIST_PREP_NO_XCD
  const int bid = blockIdx.x;
#else
  const int bid = xcd_contiguous_block(blockIdx.x, gridDim.x);
#endif
  int l = 0;
#pragma unroll
  for (int i = 1; i < LIST_N_IMG_LEVELS; ++i)
    if (i < a.n_levels && bid >= a.lv[i].wg_begin) l = i;
  const int cgroups = a.lv[l].cgroups;
  int idx = bid - a.lv[l].wg_begin;
  const int xt = idx % a.nxt; idx /= a.nxt;
  const int cg = idx % cgroups; idx /= cgroups;
  const int yt = idx % a.lv[l].nyt;
  const int b = idx / a.lv[l].nyt;
  const int RY = a.lv[l].RY;
  const ListMap2D m = a.lv[l].m;
  const int coff = a.lv[l].coff, vec = a.lv[l].vec;
  const int ms = a.ms;
  const int q = threadIdx.x & 7, xi = threadIdx.x >> 3;
  const int xo = xt * kRowsPx + xi;
  const bool active = xo < ms;
  const int x = active ? xo : ms - 1;
  const int c = cg * kRowsCg + 8 * q;

  const float sy = ms > 1 ? (float)(m.H - 1) / (float)(ms - 1) : 0.f;
  const float sx = ms > 1 ? (float)(m.W - 1) / (float)(ms - 1) : 0.f;
  const float fx = sx * (float)x;
  const int x0 = min((int)fx, m.W - 1);
  const int x1 = x0 + (x0 < m.W - 1 ? 1 : 0);
  const float wx1 = fx - (float)x0, wx0 = 1.f - wx1;
  const float* p0 = m.data + (int64_t)b * m.sb + (int64_t)c * m.sc + (int64_t)x0 * m.sw;
  const float* p1 = m.data + (int64_t)b * m.sb + (int64_t)c * m.sc + (int64_t)x1 * m.sw;
#ifndef LIST_PREP_NO_PAIR
  // NCHW-like source (x contiguous): taps x0, x1 = x0 + 1 (or x0 at the right edge) come from the pair (xb, xb + 1)
  const bool pair = !vec && m.sw == 1 && m.W >= 2;
#else
  const bool pair = false;
#endif
  const int xb = min(x0, m.W - 2);
  const bool sel0 = x0 != xb, sel1 = x1 != xb;
  const float* pb = m.data + (int64_t)b * m.sb + (int64_t)c * m.sc + (int64_t)xb;

  // (Tried: fetching the source-row segment of the tile with loads that run along x and turning it through LDS
  // into [column][64 channels] -- 0.29 ms instead of 0.19: two barriers and an LDS round trip per fetch lengthen
  // the chain of dependent fetches that bounds a workgroup.  Pixel-tile-major block order: 0.73 ms.  One workgroup
  // per 8 columns x all 1024 channels (16 KB contiguous per row): 0.22 ms.  Non-temporal stores, 8/16/32 columns
  // or rows per workgroup: within 3 %.)
  auto hrow = [&](int r, float (&h)[8]) {
    const float* r0 = p0 + (int64_t)r * m.sh;
    const float* r1 = p1 + (int64_t)r * m.sh;
    float u[8], v[8];
    if (vec) {                      // channels-last source: 8 channels are 32 contiguous bytes
      const float4 a0 = *(const float4*)r0, a1 = *(const float4*)(r0 + 4);
      const float4 b0 = *(const float4*)r1, b1 = *(const float4*)(r1 + 4);
      u[0] = a0.x; u[1] = a0.y; u[2] = a0.z; u[3] = a0.w; u[4] = a1.x; u[5] = a1.y; u[6] = a1.z; u[7] = a1.w;
      v[0] = b0.x; v[1] = b0.y; v[2] = b0.z; v[3] = b0.w; v[4] = b1.x; v[5] = b1.y; v[6] = b1.z; v[7] = b1.w;
    } else if (pair) {              // x-contiguous source: the two taps of a channel are one 8-byte load
      const float* rb = pb + (int64_t)r * m.sh;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        typedef __attribute__((ext_vector_type(2))) float f2u __attribute__((aligned(4)));
        const f2u a = *(const f2u*)(rb + (int64_t)k * m.sc);
        u[k] = sel0 ? a.y : a.x; v[k] = sel1 ? a.y : a.x;
      }
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) { u[k] = r0[(int64_t)k * m.sc]; v[k] = r1[(int64_t)k * m.sc]; }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) h[k] = u[k] * wx0 + v[k] * wx1;
  };

  const int y_first = yt * RY, y_end = min(y_first + RY, ms);
  float top[8], bot[8];
  int row_top = -1, row_bot = -1;
  int64_t oi = ((int64_t)(b * ms + y_first) * ms + xo) * a.Ct + coff + c;
  const int64_t ostep = (int64_t)ms * a.Ct;
#pragma unroll 1
  for (int y = y_first; y < y_end; ++y, oi += ostep) {
    const float fy = sy * (float)y;
    const int y0 = min((int)fy, m.H - 1);
    const int y1 = y0 + (y0 < m.H - 1 ? 1 : 0);
    const float wy1 = fy - (float)y0, wy0 = 1.f - wy1;
    if (y0 != row_top) {                       // (wave-uniform branches: y is the same for the workgroup)
      if (y0 == row_bot) {
#pragma unroll
        for (int k = 0; k < 8; ++k) top[k] = bot[k];
      } else {
        hrow(y0, top);
      }
      row_top = y0;
    }
    if (y1 != row_bot) {
      if (y1 == row_top) {
#pragma unroll
        for (int k = 0; k < 8; ++k) bot[k] = top[k];
      } else {
        hrow(y1, bot);
      }
      row_bot = y1;
    }
    float o[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = top[k] * wy0 + bot[k] * wy1;
    if (!active) continue;
    if (F16) {
#ifndef LIST_PREP_SAT_ALWAYS
      // saturate only the rows that need it: one compare per value (|x| is a free source modifier; a NaN compares
      // false and converts to NaN, +-inf compares true) instead of the 3-instruction NaN-preserving clamp per value
      // -- the kernel is bound by vector issue and the clamp was a third of its steady-state row
      bool over = false;
#pragma unroll
      for (int k = 0; k < 8; ++k) over = over || (fabsf(o[k]) > 65504.f);
      if (over) {
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = sat_h(o[k]);
      }
      const uint2 lo = half4_inrange(make_float4(o[0], o[1], o[2], o[3]));
      const uint2 hi = half4_inrange(make_float4(o[4], o[5], o[6], o[7]));
#else
      const uint2 lo = half4(make_float4(o[0], o[1], o[2], o[3]));
      const uint2 hi = half4(make_float4(o[4], o[5], o[6], o[7]));
#endif
#ifndef LIST_PREP_NO_NT          // streamed once, read again only by the 2-D gather: keep it out of the L2 working set
      __builtin_nontemporal_store((f32x4){__builtin_bit_cast(float, lo.x), __builtin_bit_cast(float, lo.y),
                                          __builtin_bit_cast(float, hi.x), __builtin_bit_cast(float, hi.y)},
                                  (f32x4*)((unsigned short*)out + oi));
#else
      *(uint4*)((unsigned short*)out + oi) = make_uint4(lo.x, lo.y, hi.x, hi.y);
#endif
    } else {
      *(float4*)((float*)out + oi) = make_float4(o[0], o[1], o[2], o[3]);
      *(float4*)((float*)out + oi + 4) = make_float4(o[4], o[5], o[6], o[7]);
    }
  }
}

static bool rows_eligible(const ListMap2D& m, int ms, int Ct, int coff) {
  return ms >= 2 && m.C >= kRowsCg && (m.C % kRowsCg) == 0 && (coff % 8) == 0 && (Ct % 8) == 0;
}

hipError_t launch_prep_img(const ListMap2D maps[LIST_N_IMG_LEVELS], int B, int map_size, int Ct,
                           int f16, void* out, hipStream_t s, int n_levels) {
  int coff = 0;
  if (n_levels <= 0) return hipSuccess;
#ifndef LIST_PREP_IMG_NO_ROWS
  {
    PrepRowsArgs a;
    a.n_levels = 0; a.B = B; a.ms = map_size; a.Ct = Ct;
    a.nxt = (map_size + kRowsPx - 1) / kRowsPx;
    int64_t wgs = 0;
    bool all = true;
    int co = 0;
    for (int i = 0; i < n_levels; ++i) {
      const ListMap2D& m = maps[i];
      if (!rows_eligible(m, map_size, Ct, co)) { all = false; break; }
      PrepRowsLevel& lv = a.lv[a.n_levels++];
      lv.m = m; lv.coff = co; lv.wg_begin = (int)wgs; lv.cgroups = m.C / kRowsCg;
      // Output rows per workgroup.  A thread's rows form a chain of dependent source-row fetches (~2 us each):
      // about five fetches per workgroup everywhere -- few rows where every output row needs new source rows
      // (down-sampling: 2 per row), many where ten output rows share one (the 14 px level)
      {
        const float sy = (float)(m.H - 1) / (float)(map_size - 1);
        int ry = (int)(3.3f / (sy > 0.05f ? sy : 0.05f) + 0.5f);
#ifdef LIST_PREP_RY_UNIFORM
        ry = LIST_PREP_RY;
#endif
        lv.RY = ry < 2 ? 2 : (ry > LIST_PREP_RY_MAX ? LIST_PREP_RY_MAX : ry);
        lv.nyt = (map_size + lv.RY - 1) / lv.RY;
      }
      lv.vec = (m.sc == 1 && (m.sw % 4) == 0 && (m.sh % 4) == 0 && (m.sb % 4) == 0 &&
                (reinterpret_cast<uintptr_t>(m.data) & 15) == 0) ? 1 : 0;
      wgs += (int64_t)B * lv.nyt * lv.cgroups * a.nxt;
      co += m.C;
    }
    if (all && wgs > 0 && wgs < 2147483647LL) {
#ifdef LIST_PREP_PER_LEVEL           // profiling only: one launch per level, so that a kernel trace shows each
      for (int i = 0; i < a.n_levels; ++i) {
        PrepRowsArgs one = a;
        one.n_levels = 1; one.lv[0] = a.lv[i]; one.lv[0].wg_begin = 0;
        const unsigned n = (unsigned)((int64_t)B * a.lv[i].nyt * a.lv[i].cgroups * a.nxt);
        if (f16) hipLaunchKernelGGL(k_prep_img_rows<1>, dim3(n), dim3(kRowsPx * 8), 0, s, one, out);
        else hipLaunchKernelGGL(k_prep_img_rows<0>, dim3(n), dim3(kRowsPx * 8), 0, s, one, out);
      }
      return hipGetLastError();
#endif
      if (f16)
        hipLaunchKernelGGL(k_prep_img_rows<1>, dim3((unsigned)wgs), dim3(kRowsPx * 8), 0, s, a, out);
      else
        hipLaunchKernelGGL(k_prep_img_rows<0>, dim3((unsigned)wgs), dim3(kRowsPx * 8), 0, s, a, out);
      return hipGetLastError();
    }
  }
#endif
  for (int i = 0; i < n_levels; ++i) {
    const ListMap2D& m = maps[i];
    hipError_t fe = hipSuccess;
    if (try_prep_img_nhwc(m, B, map_size, Ct, coff, f16, out, s, &fe) ||
        try_prep_img_tile(m, B, map_size, Ct, coff, f16, out, s, &fe)) {
      if (fe != hipSuccess) return fe;
      coff += m.C;
      continue;
    }
    dim3 grid(B * map_size, (m.C + kResizeCg - 1) / kResizeCg);
    hipLaunchKernelGGL(k_prep_img, grid, dim3(256), 0, s, m, map_size, Ct, coff, f16, out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    coff += m.C;
  }
  return hipSuccess;
}

// --------------------------------------------------------------------------------------------
// list_prep_img_proj: an encoder level [B,C,H,W] fp32 (any strides) -> the A operand of its projection, rows
// [B*H*W][C] (fp16, or fp32 for the split formats).  The levels this runs on are small (14^2 ... 56^2 pixels,
// 3 - 13 MB per batch): lanes over pixels (coalesced along W on NCHW sources), 8 channels per thread.
// --------------------------------------------------------------------------------------------
struct LevelRowsArgs { ListMap2D m[LIST_N_IMG_LEVELS]; void* out[LIST_N_IMG_LEVELS]; int blk_begin[LIST_N_IMG_LEVELS + 1]; int n, B; };

// one launch for every projected level; a workgroup = 32 pixels x 64 channels of one image: reads run along the pixels
// (NCHW sources) or along the channels (channels-last ones), the [32][64] tile turns through LDS, writes are whole
// 128-B (fp16) / 256-B rows of 64 channels.  blockIdx.x walks (level | image, pixel block, channel group)
template <int F16>
__global__ __launch_bounds__(256) void k_img_level_rows(LevelRowsArgs a) {
  __shared__ float tile[32][65];
  int l = 0;
#pragma unroll
  for (int i = 1; i < LIST_N_IMG_LEVELS; ++i)
    if (i < a.n && (int)blockIdx.x >= a.blk_begin[i]) l = i;
  const ListMap2D m = a.m[l];
  void* __restrict__ out = a.out[l];
  const int npx = m.H * m.W, cgs = m.C / 64, pbs = (npx + 31) / 32;
  int idx = (int)blockIdx.x - a.blk_begin[l];
  const int cg = idx % cgs; idx /= cgs;
  const int pb = idx % pbs;
  const int b = idx / pbs;
  const int c0 = cg * 64, p0 = pb * 32;
  {
    const int pp = threadIdx.x & 31, oc = threadIdx.x >> 5;            // pixel, channel octet
    const int px = min(p0 + pp, npx - 1);
    const int y = px / m.W, x = px - y * m.W;
    const float* src = m.data + (int64_t)b * m.sb + (int64_t)y * m.sh + (int64_t)x * m.sw + (int64_t)(c0 + 8 * oc) * m.sc;
#pragma unroll
    for (int k = 0; k < 8; ++k) tile[pp][8 * oc + k] = src[(int64_t)k * m.sc];
  }
  __syncthreads();
  const int pp = threadIdx.x >> 3, ch = threadIdx.x & 7;               // pixel, 8-channel chunk
  if (p0 + pp >= npx) return;
  float v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = tile[pp][8 * ch + k];
  const int64_t o = ((int64_t)b * npx + p0 + pp) * m.C + c0 + 8 * ch;
  if (F16) {
    const uint2 lo = half4(make_float4(v[0], v[1], v[2], v[3])), hi = half4(make_float4(v[4], v[5], v[6], v[7]));
    *(uint4*)((unsigned short*)out + o) = make_uint4(lo.x, lo.y, hi.x, hi.y);
  } else {
    *(float4*)((float*)out + o) = make_float4(v[0], v[1], v[2], v[3]);
    *(float4*)((float*)out + o + 4) = make_float4(v[4], v[5], v[6], v[7]);
  }
}

hipError_t launch_img_level_rows(const ListMap2D* maps, void* const* outs, int n, int B, int f16, hipStream_t s) {
  if (n < 1 || n > LIST_N_IMG_LEVELS || B < 1) return hipErrorInvalidValue;
  LevelRowsArgs a;
  a.n = n; a.B = B;
  int64_t blocks = 0;
  for (int i = 0; i < n; ++i) {
    const ListMap2D& m = maps[i];
    if (m.C % 64 || m.H < 1 || m.W < 1) return hipErrorInvalidValue;
    a.m[i] = m; a.out[i] = outs[i]; a.blk_begin[i] = (int)blocks;
    blocks += (int64_t)B * ((m.H * m.W + 31) / 32) * (m.C / 64);
    if (blocks >= 2147483647LL) return hipErrorInvalidValue;
  }
  for (int i = n; i < LIST_N_IMG_LEVELS; ++i) { a.m[i] = maps[0]; a.out[i] = outs[0]; a.blk_begin[i] = (int)blocks; }
  a.blk_begin[LIST_N_IMG_LEVELS] = (int)blocks;
  if (f16) hipLaunchKernelGGL(k_img_level_rows<1>, dim3((unsigned)blocks), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(k_img_level_rows<0>, dim3((unsigned)blocks), dim3(256), 0, s, a);
  return hipGetLastError();
}

// --------------------------------------------------------------------------------------------
// list_prep_img_proj: out[b][y][x][coff + n] = sum_l resize(P_l)[b][y][x][n] -- the bilinear align_corners resize of
// k_prep_img_rows (same index and weight arithmetic, same separable order, same row-streaming scheme: a thread owns
// (output column, 8 channels) and walks down its rows with the two horizontally interpolated source rows of EVERY
// source level in registers), the levels' results added in level order in fp32.  The sources are the projected
// levels, channels-last fp32 [B][H_l][W_l][H1]: 8 channels of a tap are 32 contiguous bytes.
// --------------------------------------------------------------------------------------------
constexpr int kProjMaxSrc = LIST_N_IMG_LEVELS;
struct ProjSumArgs { ListMap2D src[kProjMaxSrc]; int n_src, B, ms, Ct, coff, H1, RY, nyt, nxt; };

// NL = number of source levels (compile time: the per-level row pairs live in registers, 16 per level)
template <int F16, int NL, int SRC16>
__global__ __launch_bounds__(LIST_PREP_THREADS) void k_proj_resize_sum(ProjSumArgs a, void* __restrict__ out) {
  const int bid = xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int cgroups = a.H1 / kRowsCg;
  int idx = bid;
  const int xt = idx % a.nxt; idx /= a.nxt;
  const int cg = idx % cgroups; idx /= cgroups;
  const int yt = idx % a.nyt;
  const int b = idx / a.nyt;
  const int ms = a.ms;
  const int q = threadIdx.x & 7, xi = threadIdx.x >> 3;
  const int xo = xt * kRowsPx + xi;
  const bool active = xo < ms;
  const int x = active ? xo : ms - 1;
  const int c = cg * kRowsCg + 8 * q;

  float top[NL][8], bot[NL][8];
  int row_top[NL], row_bot[NL];
  int64_t p0[NL], p1[NL];            // element offsets of the two x taps (row 0) in the level
  float wx0[NL], wx1[NL], sy[NL];
#pragma unroll
  for (int l = 0; l < NL; ++l) {
    const ListMap2D& m = a.src[l];
    sy[l] = ms > 1 ? (float)(m.H - 1) / (float)(ms - 1) : 0.f;
    const float sx = ms > 1 ? (float)(m.W - 1) / (float)(ms - 1) : 0.f;
    const float fx = sx * (float)x;
    const int x0 = min((int)fx, m.W - 1);
    const int x1 = x0 + (x0 < m.W - 1 ? 1 : 0);
    wx1[l] = fx - (float)x0; wx0[l] = 1.f - wx1[l];
    p0[l] = (int64_t)b * m.sb + (int64_t)x0 * m.sw + c;
    p1[l] = (int64_t)b * m.sb + (int64_t)x1 * m.sw + c;
    row_top[l] = -1; row_bot[l] = -1;
  }
  // (SRC16: the projected levels are halfs -- strides and offsets of `src` count ELEMENTS of that type; one 16-B load per
  // tap instead of two)
  auto hrow = [&](int l, int r, float (&h)[8]) {
    float u[8], v[8];
    if (SRC16) {
      const unsigned short* r0 = (const unsigned short*)a.src[l].data + p0[l] + (int64_t)r * a.src[l].sh;
      const unsigned short* r1 = (const unsigned short*)a.src[l].data + p1[l] + (int64_t)r * a.src[l].sh;
      MapT<1>::unpack(*(const uint4*)r0, u);
      MapT<1>::unpack(*(const uint4*)r1, v);
    } else {
      const float* r0 = a.src[l].data + p0[l] + (int64_t)r * a.src[l].sh;
      const float* r1 = a.src[l].data + p1[l] + (int64_t)r * a.src[l].sh;
      const float4 a0 = *(const float4*)r0, a1 = *(const float4*)(r0 + 4);
      const float4 b0 = *(const float4*)r1, b1 = *(const float4*)(r1 + 4);
      u[0] = a0.x; u[1] = a0.y; u[2] = a0.z; u[3] = a0.w; u[4] = a1.x; u[5] = a1.y; u[6] = a1.z; u[7] = a1.w;
      v[0] = b0.x; v[1] = b0.y; v[2] = b0.z; v[3] = b0.w; v[4] = b1.x; v[5] = b1.y; v[6] = b1.z; v[7] = b1.w;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) h[k] = u[k] * wx0[l] + v[k] * wx1[l];
  };

  const int y_first = yt * a.RY, y_end = min(y_first + a.RY, ms);
  int64_t oi = ((int64_t)(b * ms + y_first) * ms + xo) * a.Ct + a.coff + c;
  const int64_t ostep = (int64_t)ms * a.Ct;
#pragma unroll 1
  for (int y = y_first; y < y_end; ++y, oi += ostep) {
    float o[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = 0.f;
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      const int H = a.src[l].H;
      const float fy = sy[l] * (float)y;
      const int y0 = min((int)fy, H - 1);
      const int y1 = y0 + (y0 < H - 1 ? 1 : 0);
      const float wy1 = fy - (float)y0, wy0 = 1.f - wy1;
      if (y0 != row_top[l]) {                    // (wave-uniform: y is the same for the workgroup)
        if (y0 == row_bot[l]) {
#pragma unroll
          for (int k = 0; k < 8; ++k) top[l][k] = bot[l][k];
        } else {
          hrow(l, y0, top[l]);
        }
        row_top[l] = y0;
      }
      if (y1 != row_bot[l]) {
        if (y1 == row_top[l]) {
#pragma unroll
          for (int k = 0; k < 8; ++k) bot[l][k] = top[l][k];
        } else {
          hrow(l, y1, bot[l]);
        }
        row_bot[l] = y1;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] += top[l][k] * wy0 + bot[l][k] * wy1;
    }
    if (!active) continue;
    if (F16) {
      const uint2 lo = half4(make_float4(o[0], o[1], o[2], o[3]));
      const uint2 hi = half4(make_float4(o[4], o[5], o[6], o[7]));
      *(uint4*)((unsigned short*)out + oi) = make_uint4(lo.x, lo.y, hi.x, hi.y);
    } else {
      *(float4*)((float*)out + oi) = make_float4(o[0], o[1], o[2], o[3]);
      *(float4*)((float*)out + oi + 4) = make_float4(o[4], o[5], o[6], o[7]);
    }
  }
}

hipError_t launch_proj_resize_sum(const ListMap2D* src, int n_src, int B, int map_size, int Ct, int coff, int f16,
                                  void* out, hipStream_t s, int src_f16) {
  if (n_src < 1 || n_src > kProjMaxSrc || map_size < 2 || B < 1) return hipErrorInvalidValue;
  ProjSumArgs a;
  a.n_src = n_src; a.B = B; a.ms = map_size; a.Ct = Ct; a.coff = coff; a.H1 = src[0].C;
  if (a.H1 % kRowsCg || coff % 8 || Ct % 8) return hipErrorInvalidValue;
  float sy_max = 0.f;
  for (int l = 0; l < n_src; ++l) {
    const ListMap2D& m = src[l];
    const int al = src_f16 ? 8 : 4;           // elements per 16 bytes of the sources
    if (m.C != a.H1 || m.sc != 1 || (m.sw % al) || (m.sh % al) || (m.sb % al) || (reinterpret_cast<uintptr_t>(m.data) & 15))
      return hipErrorInvalidValue;
    a.src[l] = m;
    const float sy = (float)(m.H - 1) / (float)(map_size - 1);
    sy_max = sy > sy_max ? sy : sy_max;
  }
  for (int l = n_src; l < kProjMaxSrc; ++l) a.src[l] = src[0];
  // output rows per workgroup
  // (measured at the metric's three levels, sy_max 0.40: 4 / 8 / 16 / 24 / 32 rows -> the whole prep 0.191 / 0.179 / 0.171 /
  // 0.169 / 0.175 ms: the first row of a workgroup fetches two source rows of EVERY level, so twice k_prep_img_rows's rows)
  int ry = (int)(6.6f / (sy_max > 0.05f ? sy_max : 0.05f) + 0.5f);
  a.RY = ry < 2 ? 2 : (ry > LIST_PREP_RY_MAX ? LIST_PREP_RY_MAX : ry);
  a.nyt = (map_size + a.RY - 1) / a.RY;
  a.nxt = (map_size + kRowsPx - 1) / kRowsPx;
  const int64_t wgs = (int64_t)B * a.nyt * (a.H1 / kRowsCg) * a.nxt;
  if (wgs <= 0 || wgs >= 2147483647LL) return hipErrorInvalidValue;
  const dim3 grid((unsigned)wgs), block(kRowsPx * 8);
#define LIST_PROJ_SUM(NL)                                                                      \
  case NL:                                                                                     \
    if (f16 && src_f16) hipLaunchKernelGGL((k_proj_resize_sum<1, NL, 1>), grid, block, 0, s, a, out);   \
    else if (f16) hipLaunchKernelGGL((k_proj_resize_sum<1, NL, 0>), grid, block, 0, s, a, out);         \
    else if (src_f16) return hipErrorInvalidValue;                                                       \
    else hipLaunchKernelGGL((k_proj_resize_sum<0, NL, 0>), grid, block, 0, s, a, out);                  \
    break;
  switch (n_src) { LIST_PROJ_SUM(1) LIST_PROJ_SUM(2) LIST_PROJ_SUM(3) LIST_PROJ_SUM(4) LIST_PROJ_SUM(5) default: return hipErrorInvalidValue; }
#undef LIST_PROJ_SUM
  return hipGetLastError();
}

// --------------------------------------------------------------------------------------------
// [B,C,D,H,W] (any strides) -> per image [D*H*W][C].  grid = (ceil(DHW/64), B); block = 256.
// Reads are coalesced along W (lane = voxel), writes along C through an LDS tile [C][65].
// --------------------------------------------------------------------------------------------
constexpr int kTrMaxC = 128;

__global__ __launch_bounds__(256) void k_transpose_vox(ListMap3D m, int c_begin, int nc, int f16,
                                                       void* __restrict__ out) {
  __shared__ float tile[kTrMaxC * 65];
  const int nvox = m.D * m.H * m.W;
  const int b = blockIdx.y;
  const int v0 = blockIdx.x * 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int v = v0 + lane;
  if (v < nvox) {
    const int hw = m.H * m.W;
    const int z = v / hw, r = v - z * hw;
    const int y = r / m.W, x = r - y * m.W;
    const int64_t at = (int64_t)b * m.sb + (int64_t)z * m.sd + (int64_t)y * m.sh + (int64_t)x * m.sw;
    if (m.dtype == LIST_MAP_F16) {
      const _Float16* src = (const _Float16*)m.data + at;
      for (int c = wave; c < nc; c += 4) tile[c * 65 + lane] = (float)src[(int64_t)(c_begin + c) * m.sc];
    } else {
      const float* src = (const float*)m.data + at;
      for (int c = wave; c < nc; c += 4) tile[c * 65 + lane] = src[(int64_t)(c_begin + c) * m.sc];
    }
  }
  __syncthreads();
  const int nv = min(64, nvox - v0);
  const int64_t dst = ((int64_t)b * nvox + v0) * m.C + c_begin;
  for (int idx = threadIdx.x; idx < nv * nc; idx += 256) {
    const int vl = idx / nc, c = idx - vl * nc;
    put_map(out, dst + (int64_t)vl * m.C + c, tile[c * 65 + vl], f16);
  }
}

#ifndef LIST_TR_TILE
#define LIST_TR_TILE 2048
#endif
// elements per tile.  Measured (all five levels, 2.15 GB): 16384: 0.464 ms, 8192: 0.428, 4096: 0.405, 2048: 0.353
// (6.1 TB/s, the rate of a plain copy), 1024: 0.369 -- 8-KB tiles keep 4x as many workgroups, i.e. independent
// load -> LDS -> store chains, in flight per CU as the 32-KB tiles of round 1
constexpr int kTrTile = LIST_TR_TILE;
// Fast path (spatially contiguous source, C in {16,32,64,128}): a workgroup moves a kTrTile-element
// tile = V voxels x C channels (V = kTrTile / C).  16-B global loads along the voxel axis, 16-B global
// stores along the channel axis (fully contiguous), LDS image [v][c] with the element index XORed by
// ((v>>2)&7)<<2 (a bijection inside each group of 4 voxels) so that the transposing ds_write_b32
// pattern spreads over 8 bank groups and the ds_read_b128 of 4 channels stays 16-B aligned.
template <int C, int F16>
__device__ __forceinline__ void transpose_vox_tile(const float* __restrict__ src, int64_t sb, int64_t sc, int nvox,
                                                   void* __restrict__ out, float* __restrict__ tile, int tile_x,
                                                   int b) {
  constexpr int V = kTrTile / C;          // voxels per tile
  constexpr int V4 = V / 4;
  const int v0 = tile_x * V;
  const float* in = src + (int64_t)b * sb + v0;
#pragma unroll
  for (int i = 0; i < kTrTile / 1024; ++i) {
    const int idx = threadIdx.x + 256 * i;
    const int c = idx / V4, v4 = idx % V4;
    // streamed once: non-temporal loads and stores keep the copy out of the L2 working set
    // (measured 4.3 -> 5.4 TB/s on the 128^3 x 16 level)
    const f32x4 tv = __builtin_nontemporal_load((const f32x4*)(in + (int64_t)c * sc + 4 * v4));
    const float4 t = make_float4(tv[0], tv[1], tv[2], tv[3]);
    const int swz = (v4 & 7) << 2;
    const int a = (4 * v4) * C + c;
    tile[(a) ^ swz] = t.x;
    tile[(a + C) ^ swz] = t.y;
    tile[(a + 2 * C) ^ swz] = t.z;
    tile[(a + 3 * C) ^ swz] = t.w;
  }
  __syncthreads();
  const int64_t dst = ((int64_t)b * nvox + v0) * C;
  if (F16) {
    // 8 channels (two swizzled 16-B LDS reads) -> one 16-B store of 8 halfs
#pragma unroll
    for (int i = 0; i < (kTrTile + 2047) / 2048; ++i) {
      const int idx = threadIdx.x + 256 * i;          // = v * (C/8) + c8
      if (kTrTile % 2048 && idx >= kTrTile / 8) break;
      const int v = idx / (C / 8);
      const int a = idx * 8;                           // v * C + 8 * c8
      const int sw = ((v >> 2) & 7) << 2;
      const float4 lo = *(const float4*)(tile + (a ^ sw));
      const float4 hi = *(const float4*)(tile + ((a + 4) ^ sw));
      const uint2 l = half4(lo), h = half4(hi);
      __builtin_nontemporal_store((f32x4){__builtin_bit_cast(float, l.x), __builtin_bit_cast(float, l.y),
                                          __builtin_bit_cast(float, h.x), __builtin_bit_cast(float, h.y)},
                                  (f32x4*)((unsigned short*)out + dst + a));
    }
  } else {
#pragma unroll
    for (int i = 0; i < kTrTile / 1024; ++i) {
      const int idx = threadIdx.x + 256 * i;          // = v * (C/4) + c4
      const int v = idx / (C / 4);
      const int a = idx * 4;                           // v * C + 4 * c4
      __builtin_nontemporal_store(*(const f32x4*)(tile + (a ^ (((v >> 2) & 7) << 2))),
                                  (f32x4*)((float*)out + dst + a));
    }
  }
}

template <int C, int F16>
__global__ __launch_bounds__(256) void k_transpose_vox_tile(const float* __restrict__ src, int64_t sb,
                                                            int64_t sc, int nvox,
                                                            void* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) float tile[kTrTile];
  transpose_vox_tile<C, F16>(src, sb, sc, nvox, out, tile, blockIdx.x, blockIdx.y);
}

// All levels that take the tile path in ONE launch (the five hand-offs of a step are independent: one grid fills
// the tails of the small levels with the big one's tiles and drops four dependent-launch gaps).
struct TransposeJob { const float* src; void* out; int64_t sb, sc; int nvox, C, f16, tiles_x, wg_begin; };
struct TransposeJobs { TransposeJob j[LIST_N_VOX_LEVELS]; int n; int B; };

__global__ __launch_bounds__(256) void k_transpose_vox_fused(TransposeJobs a) {
  __shared__ __attribute__((aligned(16))) float tile[kTrTile];
  int l = 0;
#pragma unroll
  for (int i = 1; i < LIST_N_VOX_LEVELS; ++i)
    if (i < a.n && (int)blockIdx.x >= a.j[i].wg_begin) l = i;
  const TransposeJob j = a.j[l];
  const int idx = blockIdx.x - j.wg_begin;
  const int tx = idx % j.tiles_x, b = idx / j.tiles_x;
#define LIST_TR_CASE(CC)                                                                          \
  case CC:                                                                                        \
    if (j.f16) transpose_vox_tile<CC, 1>(j.src, j.sb, j.sc, j.nvox, j.out, tile, tx, b);          \
    else transpose_vox_tile<CC, 0>(j.src, j.sb, j.sc, j.nvox, j.out, tile, tx, b);                \
    break;
  switch (j.C) {
    LIST_TR_CASE(16) LIST_TR_CASE(32) LIST_TR_CASE(64) LIST_TR_CASE(128)
    default: break;
  }
#undef LIST_TR_CASE
}

bool transpose_tile_eligible(const ListMap3D& m, const void* out) {
  const int nvox = m.D * m.H * m.W;
  const bool spatial_contig = m.sw == 1 && m.sh == m.W && m.sd == (int64_t)m.H * m.W;
  const bool aligned = (reinterpret_cast<uintptr_t>(m.data) & 15) == 0 && (m.sb % 4) == 0 &&
                       (m.sc % 4) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0;
  return m.dtype == LIST_MAP_F32 && spatial_contig && aligned &&
         (m.C == 16 || m.C == 32 || m.C == 64 || m.C == 128) && nvox % (kTrTile / m.C) == 0;
}

hipError_t launch_transpose_vox_fused(const ListMap3D* maps, void* const* outs, const int* f16, int n, int B,
                                      hipStream_t s) {
  TransposeJobs a;
  a.n = 0; a.B = B;
  int64_t wgs = 0;
  for (int i = 0; i < n; ++i) {
    const ListMap3D& m = maps[i];
    TransposeJob& j = a.j[a.n++];
    j.src = (const float*)m.data; j.out = outs[i]; j.sb = m.sb; j.sc = m.sc; j.nvox = m.D * m.H * m.W; j.C = m.C;
    j.f16 = f16[i]; j.tiles_x = j.nvox / (kTrTile / m.C); j.wg_begin = (int)wgs;
    wgs += (int64_t)j.tiles_x * B;
  }
  if (wgs <= 0 || wgs >= 2147483647LL) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_transpose_vox_fused, dim3((unsigned)wgs), dim3(256), 0, s, a);
  return hipGetLastError();
}

template <int C>
static hipError_t launch_transpose_tile(const ListMap3D& m, int B, int f16, void* out, hipStream_t s) {
  const int nvox = m.D * m.H * m.W;
  const dim3 grid(nvox / (kTrTile / C), B);
  if (f16)
    hipLaunchKernelGGL((k_transpose_vox_tile<C, 1>), grid, dim3(256), 0, s, (const float*)m.data, m.sb, m.sc, nvox, out);
  else
    hipLaunchKernelGGL((k_transpose_vox_tile<C, 0>), grid, dim3(256), 0, s, (const float*)m.data, m.sb, m.sc, nvox, out);
  return hipGetLastError();
}

hipError_t launch_transpose_vox(const ListMap3D& m, int B, int f16, void* out, hipStream_t s) {
  const int nvox = m.D * m.H * m.W;
  const bool spatial_contig = m.sw == 1 && m.sh == m.W && m.sd == (int64_t)m.H * m.W;
  const bool aligned = (reinterpret_cast<uintptr_t>(m.data) & 15) == 0 && (m.sb % 4) == 0 &&
                       (m.sc % 4) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0;
  if (m.dtype == LIST_MAP_F32 && spatial_contig && aligned && (m.C == 16 || m.C == 32 || m.C == 64 || m.C == 128) &&
      nvox % (kTrTile / m.C) == 0) {
    switch (m.C) {
      case 16: return launch_transpose_tile<16>(m, B, f16, out, s);
      case 32: return launch_transpose_tile<32>(m, B, f16, out, s);
      case 64: return launch_transpose_tile<64>(m, B, f16, out, s);
      default: return launch_transpose_tile<128>(m, B, f16, out, s);
    }
  }
  for (int c0 = 0; c0 < m.C; c0 += kTrMaxC) {
    const int nc = m.C - c0 < kTrMaxC ? m.C - c0 : kTrMaxC;
    dim3 grid((nvox + 63) / 64, B);
    hipLaunchKernelGGL(k_transpose_vox, grid, dim3(256), 0, s, m, c0, nc, f16, out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

// --------------------------------------------------------------------------------------------
// MLP parameter repack
// --------------------------------------------------------------------------------------------
__device__ __forceinline__ void prep_w0_elem(int64_t i, const float* __restrict__ w0, const FeatLayout& L, int H1,
                                             int fmt, unsigned short* __restrict__ hi) {
  if (i >= (int64_t)H1 * L.Kp) return;
  const int n = (int)(i / L.Kp), kp = (int)(i - (int64_t)n * L.Kp);
  const int kr = ref_index_of(L, kp);
  const float v = kr >= 0 ? w0[(int64_t)n * L.F + kr] : 0.f;
  if (fmt == FMT_FP16) { hi[i] = f2h(v); return; }
  const unsigned short h = f2bf(v);
  hi[xi_off(i)] = h;                          // split formats: hi / lo halfs interleaved in 64-B blocks (xi_off)
  hi[xi_off(i) + kXiLo] = bf_lo(v, h);
}

// blocks `blk` of `nblk` (256 threads each) over n4 float4s
__device__ __forceinline__ void split_range(int blk, int nblk, const float4* __restrict__ x, uint2* __restrict__ hi,
                                            uint2* __restrict__ lo, int64_t n4, int fmt) {
  for (int64_t i = (int64_t)blk * 256 + threadIdx.x; i < n4; i += (int64_t)nblk * 256) {
    if (fmt == FMT_FP16) { hi[i] = half4(x[i]); continue; }
    uint2 h, l;
    split4(x[i], h, l);
    hi[i] = h;
    if (lo) lo[i] = l;
  }
}

__global__ __launch_bounds__(256) void k_split(const float4* __restrict__ x, uint2* __restrict__ hi,
                                               uint2* __restrict__ lo, int64_t n4, int fmt) {
  split_range(blockIdx.x, gridDim.x, x, hi, lo, n4, fmt);
}

// fp32 [rows][C] (C % 32 == 0) -> bf16 hi / lo halfs interleaved in 64-B blocks (xi_off): the A operand of the
// perceptual-map projection in the split formats
__global__ __launch_bounds__(256) void k_split_xi(const float4* __restrict__ x, unsigned short* __restrict__ out,
                                                  int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    uint2 h, l;
    split4(x[i], h, l);
    unsigned short* d = out + xi_off(4 * i);
    *(uint2*)d = h;
    *(uint2*)(d + kXiLo) = l;
  }
}

hipError_t launch_split_xi(const float* x, unsigned short* out, int64_t n, hipStream_t s) {
  const int64_t n4 = n / 4;
  int64_t blocks = (n4 + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_split_xi, dim3((unsigned)blocks), dim3(256), 0, s, (const float4*)x, out, n4);
  return hipGetLastError();
}

hipError_t launch_split(const float* x, unsigned short* hi, unsigned short* lo, int64_t n, int fmt,
                        hipStream_t s, int order) {
  const int64_t n4 = n / 4;
  int64_t blocks = (n4 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (blocks < 1) blocks = 1;
  LIST_LAUNCH(k_split, dim3((unsigned)blocks), dim3(256), 0, s, order, (const float4*)x, (uint2*)hi,
              (uint2*)lo, n4, fmt);
  return hipGetLastError();
}

// The whole repack of list_prep_mlp_weights as ONE launch (four until round 3): block ranges [0, nb0) permute / pad /
// convert W0 into gather order, [nb0, +nb1) and [.., +nb2) convert W1 and W2, the last block(s) copy the five short
// fp32 vectors (biases, w3, b3).  Four disjoint parts of `packed`, no block reads what another writes.
struct PrepWeightsArgs {
  const float *w0, *w1, *w2, *b0, *b1, *b2, *w3, *b3;
  FeatLayout L;
  int H1, H2, H3, fmt;
  int nb0, nb1, nb2;
  int64_t n4_1, n4_2;
  unsigned short *w0_hi, *w1_hi, *w1_lo, *w2_hi, *w2_lo;
  float *o0, *o1, *o2, *o3, *o4;
};

__global__ __launch_bounds__(256) void k_prep_weights(PrepWeightsArgs a) {
  int blk = blockIdx.x;
  if (blk < a.nb0) { prep_w0_elem((int64_t)blk * 256 + threadIdx.x, a.w0, a.L, a.H1, a.fmt, a.w0_hi); return; }
  blk -= a.nb0;
  if (blk < a.nb1) { split_range(blk, a.nb1, (const float4*)a.w1, (uint2*)a.w1_hi, (uint2*)a.w1_lo, a.n4_1, a.fmt); return; }
  blk -= a.nb1;
  if (blk < a.nb2) { split_range(blk, a.nb2, (const float4*)a.w2, (uint2*)a.w2_hi, (uint2*)a.w2_lo, a.n4_2, a.fmt); return; }
  blk -= a.nb2;
  int i = blk * 256 + threadIdx.x;
  if (i < a.H1) { a.o0[i] = a.b0[i]; return; }
  i -= a.H1;
  if (i < a.H2) { a.o1[i] = a.b1[i]; return; }
  i -= a.H2;
  if (i < a.H3) { a.o2[i] = a.b2[i]; return; }
  i -= a.H3;
  if (i < a.H3) { a.o3[i] = a.w3[i]; return; }
  i -= a.H3;
  if (i == 0) a.o4[0] = a.b3[0];
}

static int split_blocks(int64_t n4) {
  int64_t b = (n4 + 255) / 256;
  return (int)(b > 8192 ? 8192 : b < 1 ? 1 : b);
}

hipError_t launch_prep_weights(const ListMlpWeights& w, const FeatLayout& L, const PackedMlp& P,
                               char* packed, hipStream_t s) {
  const int64_t n0 = (int64_t)w.H1 * L.Kp;
  PrepWeightsArgs a;
  a.w0 = w.w0; a.w1 = w.w1; a.w2 = w.w2; a.b0 = w.b0; a.b1 = w.b1; a.b2 = w.b2; a.w3 = w.w3; a.b3 = w.b3;
  a.L = L; a.H1 = w.H1; a.H2 = w.H2; a.H3 = w.H3;
  a.fmt = w.precision == LIST_PREC_FP16 ? FMT_FP16 : FMT_BF16_SPLIT;
  a.n4_1 = (int64_t)w.H2 * w.H1 / 4;
  a.n4_2 = (int64_t)w.H3 * w.H2 / 4;
  a.nb0 = (int)((n0 + 255) / 256);
  a.nb1 = split_blocks(a.n4_1);
  a.nb2 = split_blocks(a.n4_2);
  const int nbs = (w.H1 + w.H2 + 2 * w.H3 + 1 + 255) / 256;
  a.w0_hi = (unsigned short*)(packed + P.w0_hi);
  a.w1_hi = (unsigned short*)(packed + P.w1_hi); a.w1_lo = (unsigned short*)(packed + P.w1_lo);
  a.w2_hi = (unsigned short*)(packed + P.w2_hi); a.w2_lo = (unsigned short*)(packed + P.w2_lo);
  a.o0 = (float*)(packed + P.b0); a.o1 = (float*)(packed + P.b1); a.o2 = (float*)(packed + P.b2);
  a.o3 = (float*)(packed + P.w3); a.o4 = (float*)(packed + P.b3);
  // (in stream order: the first kernel behind whatever produced the weights)
  hipLaunchKernelGGL(k_prep_weights, dim3((unsigned)(a.nb0 + a.nb1 + a.nb2 + nbs)), dim3(256), 0, s, a);
  return hipGetLastError();
}

}  // namespace list
