// Map element formats and the per-tap interpolation arithmetic of the gathers (gather_kernels.hip and the fused
// fc_0 kernel, fused_fc0_kernels.hip, which must produce the SAME bits as the stand-alone 2-D gather).
#pragma once

#include "list_common.h"

namespace list {

// ---- map element formats -------------------------------------------------------------------------------
// A lane owns V consecutive channels of a channels-last map: 4 floats (fp32 maps) or 8 halfs (fp16
// maps); either way one 16-B load per tap.  All interpolation arithmetic is fp32.
template <int F16> struct MapT;
template <> struct MapT<0> {
  static constexpr int V = 4;
  using Raw = float4;
  static __device__ __forceinline__ Raw load(const void* base, int64_t off) {
    return *(const float4*)((const float*)base + off);
  }
  static __device__ __forceinline__ void unpack(const Raw& r, float (&f)[4]) {
    f[0] = r.x; f[1] = r.y; f[2] = r.z; f[3] = r.w;
  }
};
template <> struct MapT<1> {
  static constexpr int V = 8;
  using Raw = uint4;
  static __device__ __forceinline__ Raw load(const void* base, int64_t off) {
    return *(const uint4*)((const unsigned short*)base + off);
  }
  static __device__ __forceinline__ void unpack(const Raw& r, float (&f)[8]) {
    f[0] = h2f((unsigned short)(r.x & 0xffff)); f[1] = h2f((unsigned short)(r.x >> 16));
    f[2] = h2f((unsigned short)(r.y & 0xffff)); f[3] = h2f((unsigned short)(r.y >> 16));
    f[4] = h2f((unsigned short)(r.z & 0xffff)); f[5] = h2f((unsigned short)(r.z >> 16));
    f[6] = h2f((unsigned short)(r.w & 0xffff)); f[7] = h2f((unsigned short)(r.w >> 16));
  }
};

template <typename M>
__device__ __forceinline__ void tap_mul(const typename M::Raw& r, float w, float (&a)[M::V]) {
  float f[M::V];
  M::unpack(r, f);
#pragma unroll
  for (int c = 0; c < M::V; ++c) a[c] = f[c] * w;
}
template <typename M>
__device__ __forceinline__ void tap_fma(const typename M::Raw& r, float w, float (&a)[M::V]) {
  float f[M::V];
  M::unpack(r, f);
#pragma unroll
  for (int c = 0; c < M::V; ++c) a[c] = fmaf(f[c], w, a[c]);
}

}  // namespace list
