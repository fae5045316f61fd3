// Host-only sanitizer build of the C ABI (tests/test_capi_sanitizers.py): list_capi.hip's argument validation,
// workspace carving and chunk arithmetic run under AddressSanitizer + UBSan on the CPU.  Every kernel launcher the
// C ABI calls is replaced by a stub that aborts: the checks exercised here return BEFORE any launch, so reaching a stub
// is a test failure (a validation hole), not a skipped device call.
#include <stdio.h>
#include <stdlib.h>

#include "list_common.h"

namespace list {
[[noreturn]] static void reached(const char* what) {
  fprintf(stderr, "capi_host_asan: launcher %s reached -- a validation path let a call through\n", what);
  abort();
}
#define STUB(sig, name) sig { reached(name); }
STUB(hipError_t launch_prep_img(const ListMap2D*, int, int, int, int, void*, hipStream_t, int), "launch_prep_img")
STUB(hipError_t launch_img_level_rows(const ListMap2D*, void* const*, int, int, int, hipStream_t), "launch_img_level_rows")
STUB(hipError_t launch_proj_resize_sum(const ListMap2D*, int, int, int, int, int, int, void*, hipStream_t, int), "launch_proj_resize_sum")
STUB(hipError_t launch_transpose_vox(const ListMap3D&, int, int, void*, hipStream_t), "launch_transpose_vox")
bool transpose_tile_eligible(const ListMap3D&, const void*) { return false; }
STUB(hipError_t launch_transpose_vox_fused(const ListMap3D*, void* const*, const int*, int, int, hipStream_t), "launch_transpose_vox_fused")
STUB(hipError_t launch_prep_weights(const ListMlpWeights&, const FeatLayout&, const PackedMlp&, char*, hipStream_t), "launch_prep_weights")
STUB(hipError_t launch_split_xi(const float*, unsigned short*, int64_t, hipStream_t), "launch_split_xi")
STUB(hipError_t launch_split(const float*, unsigned short*, unsigned short*, int64_t, int, hipStream_t, int), "launch_split")
STUB(hipError_t launch_sort_points(const GatherParams&, const ListQueryArgs&, const SortBuffers&, hipStream_t), "launch_sort_points")
STUB(hipError_t launch_gather(const GatherParams&, const FeatLayout&, const ListQueryArgs&, int*, hipStream_t, bool), "launch_gather")
bool fused_fc0_eligible(const GemmParams&, int, int) { return true; }
STUB(hipError_t launch_fc0_fused(const FusedFc0Params&, int, hipStream_t), "launch_fc0_fused")
STUB(hipError_t launch_features_out(const GatherParams&, const FeatLayout&, float*, int*, hipStream_t), "launch_features_out")
// (host logic, not a launcher: the plan's box-level mask is exercised with the real predicate's shape rules)
bool gather_box_eligible(const GatherParams& g, const ListVoxLevel& lv, int col_off) {
  if (g.fmt != FMT_FP16 || lv.dtype != LIST_MAP_F16 || lv.C != 128) return false;
  if ((col_off % 8) != 0 || (g.Kp % 8) != 0 || (lv.image_stride % 8) != 0) return false;
  if (lv.W > 31 || lv.H > 31 || lv.D > 31) return false;
  return (g.rows % 64) == 0;
}
int gather_box_levels(const GatherParams& g, const FeatLayout& L, const ListQueryArgs& a) {
  int mask = 0;
  for (int l = 0; l < LIST_N_VOX_LEVELS; ++l) {
    const ListVoxLevel& lv = a.vox[l];
    const int big = lv.W > lv.H ? (lv.W > lv.D ? lv.W : lv.D) : (lv.H > lv.D ? lv.H : lv.D);
    if (kDisp * 0.5f * (float)(big - 1) < 0.99f && lv.C >= 16 && gather_box_eligible(g, lv, L.vox_off[l])) mask |= 1 << l;
  }
  return mask;
}
STUB(hipError_t launch_gather_vox_box(const GatherParams&, const ListVoxLevel&, int, hipStream_t, int), "launch_gather_vox_box")
STUB(hipError_t launch_gather_fixup(const GatherParams&, const FeatLayout&, const ListQueryArgs&, const int*, hipStream_t), "launch_gather_fixup")
STUB(hipError_t launch_percep_pool(const ListPoolArgs&, hipStream_t), "launch_percep_pool")
STUB(hipError_t launch_gemm(const GemmParams&, int, int, hipStream_t), "launch_gemm")
STUB(hipError_t launch_mlp_tail(const GemmParams&, const char*, const float*, const float*, const float*, float*, const int*, int, hipStream_t), "launch_mlp_tail")
STUB(hipError_t launch_prep_weights_bwd(const ListMlpWeights&, const FeatLayout&, const PackedMlpBwd&, char*, hipStream_t), "launch_prep_weights_bwd")
STUB(hipError_t launch_gemm_tn(const GemmTnParams&, int, hipStream_t), "launch_gemm_tn")
int wgrad_splits(int M, int N, int P, int terms) {
  const int nk = P / (terms == 3 ? 32 : 64);
  const int s = wgrad_nominal_splits(M, N);
  return s > nk ? (nk < 1 ? 1 : nk) : s;
}
STUB(hipError_t launch_wgrad_reduce(const float*, int, int, int, int, const FeatLayout*, const float*, float*, int, hipStream_t), "launch_wgrad_reduce")
STUB(hipError_t launch_grad_scale(const float*, int64_t, int, float*, float*, float*, hipStream_t), "launch_grad_scale")
STUB(hipError_t launch_head(const float*, const int*, int, int, int, const unsigned short*, const float*, const float*, unsigned short*, unsigned short*, int, hipStream_t), "launch_head")
STUB(hipError_t launch_colsum(const unsigned short*, const unsigned short*, int, int, int, int, const float*, const int*, const float*, int, float*, float*, hipStream_t), "launch_colsum")
STUB(hipError_t launch_scatter_vox(const ScatterParams&, const FeatLayout&, const ListQueryArgs&, const ListVoxLevel*, const VoxGatherBuffers&, const ScatterStreams&), "launch_scatter_vox")
STUB(hipError_t launch_img_grad(const ScatterParams&, const FeatLayout&, const ListQueryArgs&, const int*, int, void*, float*, int, float*, void* const*, hipStream_t, void*, size_t), "launch_img_grad")
STUB(hipError_t launch_rows_to_grad(const ScatterParams&, int, int, int, int*, float*, int64_t, int64_t, int64_t, hipStream_t), "launch_rows_to_grad")
STUB(hipError_t launch_grad_to_rows(const float*, int64_t, int64_t, int64_t, int, int, int, float*, float*, hipStream_t), "launch_grad_to_rows")
STUB(hipError_t launch_img_grad_to_levels(const float*, int, int, int, const ListMap2D*, hipStream_t, int, const float*), "launch_img_grad_to_levels")
}  // namespace list
