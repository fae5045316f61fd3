// Producer-side hand-off kernels (HBM-bound, no MFMA):
//   * 2-D maps: bilinear resize to map_size^2 (align_corners=True) fused with NCHW -> NHWC and
//     the channel concatenation  (replaces F.interpolate x5, network/modules.py:26-35)
//   * 3-D maps: NCDHW -> NDHWC transpose through LDS
//   * MLP parameters: column permutation + K padding + bf16 hi/lo split
#include "list_common.h"

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

__global__ __launch_bounds__(256) void k_prep_img(ListMap2D m, int ms, int Ct, int coff,
                                                  float* __restrict__ out) {
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
  float* orow = out + ((int64_t)(b * ms + y) * ms) * Ct + coff + c0;
  for (int idx = threadIdx.x; idx < ms * kResizeCg; idx += 256) {
    const int x = idx / kResizeCg;
    const int cl = idx - x * kResizeCg;
    if (cl < nc) orow[(int64_t)x * Ct + cl] = tile[x * (kResizeCg + 1) + cl];
  }
}

hipError_t launch_prep_img(const ListMap2D maps[LIST_N_IMG_LEVELS], int B, int map_size, int Ct,
                           float* out, hipStream_t s) {
  int coff = 0;
  for (int i = 0; i < LIST_N_IMG_LEVELS; ++i) {
    const ListMap2D& m = maps[i];
    dim3 grid(B * map_size, (m.C + kResizeCg - 1) / kResizeCg);
    hipLaunchKernelGGL(k_prep_img, grid, dim3(256), 0, s, m, map_size, Ct, coff, out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    coff += m.C;
  }
  return hipSuccess;
}

// --------------------------------------------------------------------------------------------
// [B,C,D,H,W] (any strides) -> per image [D*H*W][C].  grid = (ceil(DHW/64), B); block = 256.
// Reads are coalesced along W (lane = voxel), writes along C through an LDS tile [C][65].
// --------------------------------------------------------------------------------------------
constexpr int kTrMaxC = 128;

__global__ __launch_bounds__(256) void k_transpose_vox(ListMap3D m, int c_begin, int nc,
                                                       float* __restrict__ out) {
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
    const float* src = m.data + (int64_t)b * m.sb + (int64_t)z * m.sd + (int64_t)y * m.sh
                       + (int64_t)x * m.sw;
    for (int c = wave; c < nc; c += 4) tile[c * 65 + lane] = src[(int64_t)(c_begin + c) * m.sc];
  }
  __syncthreads();
  const int nv = min(64, nvox - v0);
  float* dst = out + ((int64_t)b * nvox + v0) * m.C + c_begin;
  for (int idx = threadIdx.x; idx < nv * nc; idx += 256) {
    const int vl = idx / nc, c = idx - vl * nc;
    dst[(int64_t)vl * m.C + c] = tile[c * 65 + vl];
  }
}

hipError_t launch_transpose_vox(const ListMap3D& m, int B, float* out, hipStream_t s) {
  const int nvox = m.D * m.H * m.W;
  for (int c0 = 0; c0 < m.C; c0 += kTrMaxC) {
    const int nc = m.C - c0 < kTrMaxC ? m.C - c0 : kTrMaxC;
    dim3 grid((nvox + 63) / 64, B);
    hipLaunchKernelGGL(k_transpose_vox, grid, dim3(256), 0, s, m, c0, nc, out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

// --------------------------------------------------------------------------------------------
// MLP parameter repack
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_prep_w0(const float* __restrict__ w0, FeatLayout L, int H1,
                                                 unsigned short* __restrict__ hi,
                                                 unsigned short* __restrict__ lo) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)H1 * L.Kp) return;
  const int n = (int)(i / L.Kp), kp = (int)(i - (int64_t)n * L.Kp);
  const int kr = ref_index_of(L, kp);
  const float v = kr >= 0 ? w0[(int64_t)n * L.F + kr] : 0.f;
  const unsigned short h = f2bf(v);
  hi[i] = h;
  lo[i] = f2bf(v - bf2f(h));
}

__global__ __launch_bounds__(256) void k_split(const float4* __restrict__ x, uint2* __restrict__ hi,
                                               uint2* __restrict__ lo, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    uint2 h, l;
    split4(x[i], h, l);
    hi[i] = h;
    if (lo) lo[i] = l;
  }
}

hipError_t launch_split(const float* x, unsigned short* hi, unsigned short* lo, int64_t n,
                        hipStream_t s) {
  const int64_t n4 = n / 4;
  int64_t blocks = (n4 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_split, dim3((unsigned)blocks), dim3(256), 0, s, (const float4*)x, (uint2*)hi,
                     (uint2*)lo, n4);
  return hipGetLastError();
}

hipError_t launch_prep_weights(const ListMlpWeights& w, const FeatLayout& L, const PackedMlp& P,
                               char* packed, hipStream_t s) {
  const int64_t n0 = (int64_t)w.H1 * L.Kp;
  hipLaunchKernelGGL(k_prep_w0, dim3((unsigned)((n0 + 255) / 256)), dim3(256), 0, s, w.w0, L, w.H1,
                     (unsigned short*)(packed + P.w0_hi), (unsigned short*)(packed + P.w0_lo));
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  e = launch_split(w.w1, (unsigned short*)(packed + P.w1_hi), (unsigned short*)(packed + P.w1_lo),
                   (int64_t)w.H2 * w.H1, s);
  if (e != hipSuccess) return e;
  e = launch_split(w.w2, (unsigned short*)(packed + P.w2_hi), (unsigned short*)(packed + P.w2_lo),
                   (int64_t)w.H3 * w.H2, s);
  if (e != hipSuccess) return e;
  const struct { size_t off; const float* src; size_t n; } cp[] = {
      {P.b0, w.b0, (size_t)w.H1}, {P.b1, w.b1, (size_t)w.H2}, {P.b2, w.b2, (size_t)w.H3},
      {P.w3, w.w3, (size_t)w.H3}, {P.b3, w.b3, 1}};
  for (const auto& c : cp) {
    e = hipMemcpyAsync(packed + c.off, c.src, c.n * sizeof(float), hipMemcpyDeviceToDevice, s);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

}  // namespace list
