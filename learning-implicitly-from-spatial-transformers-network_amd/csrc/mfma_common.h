// Pieces shared by the MFMA kernels (gemm_kernels.hip, fused_fc0_kernels.hip): counted vector-memory waits, LDS-DMA
// staging loads, the 16x16x32 MFMA wrapper, the 16-bit output store of the staged epilogues.
#pragma once

#include "list_common.h"

namespace list {

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
#ifdef LIST_GEMM_NO_VMWAIT   // ablation (wrong results): loads are issued but their landing is never waited for
  if (N != 0) return;
#endif
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ void glds16(const char* g, char* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
// the same with the non-temporal cache policy (aux = 2): the A stream of the forward's hidden layers (X into fc_0, H1
// into fc_1) -- rows that at most two N-tiles of one XCD read, back to back, and nobody afterwards.  Round 3, four
// interleaved pairs on one device: fc_0 0.497 -> 0.492 ms, step -0.012 ms.  (Weights, and the A operand of the
// backward's dX -- fifteen N-tiles per row --, keep the default policy.)
__device__ __forceinline__ void glds16_nt(const char* g, char* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 2);
}


constexpr int kStageLd = 68;                       // floats per staged row: 64 + 4 (row r starts 4 banks on)

template <int FP16>
__device__ __forceinline__ void store8_planes(unsigned short* __restrict__ hi, unsigned short* __restrict__ lo,
                                              int64_t off, const float (&v)[8]) {
  unsigned h[4], l[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if (FP16) {
      h[e] = f2h2(v[2 * e], v[2 * e + 1]);
    } else {
      const unsigned short h0 = f2bf(v[2 * e]), h1 = f2bf(v[2 * e + 1]);
      h[e] = (unsigned)h0 | ((unsigned)h1 << 16);
      l[e] = (unsigned)bf_lo(v[2 * e], h0) | ((unsigned)bf_lo(v[2 * e + 1], h1) << 16);
    }
  }
  *(uint4*)(hi + off) = make_uint4(h[0], h[1], h[2], h[3]);
  if (!FP16 && lo) *(uint4*)(lo + off) = make_uint4(l[0], l[1], l[2], l[3]);
}


typedef __attribute__((ext_vector_type(4))) float f32x4v;

template <int FP16>
__device__ __forceinline__ f32x4v mfma16(const bf16x8& a, const bf16x8& b, const f32x4v& c) {
  if (FP16)
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b),
                                                  c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}


}  // namespace list
