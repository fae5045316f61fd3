// Driver of the host-only sanitizer build (tests/test_capi_sanitizers.py): calls the C ABI's validation, sizing and
// chunking paths with hostile and boundary arguments -- NULLs, zero / negative / 2^31-boundary sizes, the 256^3 grid,
// misaligned buffers, unsupported channel counts -- under AddressSanitizer + UBSan.  Every call here must return an
// error code or a size WITHOUT reaching a kernel launcher (capi_host_stubs.cpp aborts there).  Device pointers are
// fake non-NULL addresses: the host side never dereferences them.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "list_hip.h"

static int g_fail = 0;
#define EXPECT(cond) do { if (!(cond)) { fprintf(stderr, "FAIL %s:%d: %s   [%s]\n", __FILE__, __LINE__, #cond, list_last_error()); ++g_fail; } } while (0)

static void* fake(uintptr_t off) { return (void*)((uintptr_t)0x7f0000000000ull + off); }     // 16-byte aligned "device" address

static ListQueryArgs good_query(int B, int N, void* ws, size_t ws_bytes) {
  ListQueryArgs a;
  memset(&a, 0, sizeof(a));
  a.B = B; a.N = N; a.query = (const float*)fake(0); a.q_sb = (int64_t)N * 3; a.q_sn = 3; a.q_sc = 1;
  a.perm[0] = 2; a.perm[1] = 1; a.perm[2] = 0; a.scale = 2.f;
  a.trans_mat = (const float*)fake(0x1000); a.img_map = fake(0x2000); a.img_dtype = LIST_MAP_F16;
  a.map_size = 137; a.img_C = 1024; a.clamp_hi = 136.f;
  const int C[6] = {1, 16, 32, 64, 128, 128}, R[6] = {128, 128, 64, 32, 16, 8};
  for (int l = 0; l < 6; ++l) {
    a.vox[l].data = fake(0x100000 * (l + 1)); a.vox[l].C = C[l]; a.vox[l].D = a.vox[l].H = a.vox[l].W = R[l];
    a.vox[l].image_stride = (int64_t)C[l] * R[l] * R[l] * R[l]; a.vox[l].dtype = C[l] == 1 ? LIST_MAP_F32 : LIST_MAP_F16;
  }
  a.packed_mlp = fake(0x9000000); a.F = 3610; a.H1 = 512; a.H2 = 256; a.H3 = 256;
  a.sdf = (float*)fake(0xA000000); a.workspace = ws; a.workspace_bytes = ws_bytes; a.precision = LIST_PREC_FP16;
  a.no_activations = 1;
  return a;
}

int main() {
  EXPECT(list_abi_version() == LIST_ABI_VERSION);
  // ---- sizes: the metric shape, the 256^3 grid, the 2^31 boundary, nonsense
  const size_t ws160k = list_query_workspace_bytes(160000, 3610, 512, 256, 256);
  EXPECT(ws160k > (size_t)2 * 160000 * 3648 * 2);
  EXPECT(list_query_chunk_rows(ws160k, 160000, 3610, 512, 256, 256) == 160000);
  const int64_t grid = (int64_t)256 * 256 * 256;
  const size_t ws_grid = list_query_workspace_bytes(grid, 3610, 512, 256, 256);
  EXPECT(list_query_chunk_rows(ws_grid, grid, 3610, 512, 256, 256) == 262144);
  EXPECT(list_query_workspace_bytes(((int64_t)1 << 31) - 1, 3610, 512, 256, 256) == ws_grid);     // capped at the chunk size
  EXPECT(list_query_workspace_bytes((int64_t)1 << 31, 3610, 512, 256, 256) == ws_grid);
  EXPECT(list_query_workspace_bytes(((int64_t)1 << 31) + 1, 3610, 512, 256, 256) == ws_grid);
  EXPECT(list_query_workspace_bytes(INT64_MAX, 3610, 512, 256, 256) == ws_grid);
  EXPECT(list_query_workspace_bytes(0, 3610, 512, 256, 256) == 0);
  EXPECT(list_query_workspace_bytes(-5, 3610, 512, 256, 256) == 0);
  EXPECT(list_query_workspace_bytes(1, 0, 512, 256, 256) == 0);
  EXPECT(list_query_chunk_rows(0, grid, 3610, 512, 256, 256) == 0);
  EXPECT(list_query_chunk_rows(1 << 20, grid, 3610, 512, 256, 256) == 0);
  EXPECT(list_query_chunk_rows(SIZE_MAX, ((int64_t)1 << 31) + 7, 3610, 512, 256, 256) == 262144);
  EXPECT(list_query_chunk_rows(ws160k, 1, 3610, 512, 256, 256) == 256);
  for (size_t b = ws160k / 2 - 3; b < ws160k / 2 + 3; ++b) {                // around an arbitrary byte count: a multiple of 256
    const int64_t r = list_query_chunk_rows(b, 160000, 3610, 512, 256, 256);
    EXPECT(r > 0 && r % 256 == 0 && r < 160000);
  }
  EXPECT(list_query_bwd_workspace_bytes(160000, 3610, 512, 256, 256, LIST_PREC_FP16) > 0);
  EXPECT(list_query_bwd_workspace_bytes(262145, 3610, 512, 256, 256, LIST_PREC_FP16) == 0);      // one chunk at most
  EXPECT(list_query_bwd_workspace_bytes(-1, 3610, 512, 256, 256, 0) == 0);
  EXPECT(list_percep_proj_bytes(1, 137, 512, LIST_PREC_FP16) == (size_t)18944 * 512 * 2);
  EXPECT(list_percep_proj_bytes(0, 137, 512, 0) == 0);
  EXPECT(list_percep_proj_scratch_bytes(8, 137, 1024, LIST_PREC_BF16X3) == (size_t)8 * 18769 * 1024 * 4);
  EXPECT(list_percep_pool_bwd_workspace_bytes(0, 1024) == 0 || true);

  // ---- query: NULLs, empty, hostile structs
  EXPECT(list_sdf_query_fwd(NULL, NULL) == LIST_ERR_ARG);
  ListQueryArgs a = good_query(8, 20000, fake(0x40000000), ws160k);
  ListQueryArgs e = a; e.B = 0;
  EXPECT(list_sdf_query_fwd(&e, NULL) == LIST_OK);                          // empty query: nothing to do
  e = a; e.N = -1;
  EXPECT(list_sdf_query_fwd(&e, NULL) == LIST_ERR_SHAPE);
  e = a; e.query = NULL;
  EXPECT(list_sdf_query_fwd(&e, NULL) == LIST_ERR_ARG);
  e = a; e.perm[1] = 3;
  EXPECT(list_sdf_query_fwd(&e, NULL) == LIST_ERR_ARG);
  e = a; e.vox[2].D = 0;
  EXPECT(list_sdf_query_fwd(&e, NULL) == LIST_ERR_SHAPE);
  e = a; e.vox[1].D = 2048; e.vox[1].H = 1024; e.vox[1].W = 1024;          // 2^31 voxels x 16 channels
  EXPECT(list_sdf_query_fwd(&e, NULL) == LIST_ERR_SHAPE);
  e = a; e.vox[1].D = 512; e.vox[1].H = 512; e.vox[1].W = 512;             // exactly 2^31 elements
  EXPECT(list_sdf_query_fwd(&e, NULL) == LIST_ERR_SHAPE);
  e = a; e.vox[3].C = 48;
  EXPECT(list_sdf_query_fwd(&e, NULL) == LIST_ERR_UNSUPPORTED);
  e = a; e.vox[3].dtype = 7;
  EXPECT(list_sdf_query_fwd(&e, NULL) == LIST_ERR_ARG);
  e = a; e.vox[4].data = (char*)e.vox[4].data + 2;                           // misaligned level
  EXPECT(list_sdf_query_fwd(&e, NULL) == LIST_ERR_SHAPE);
  e = a; e.vox[0].dtype = LIST_MAP_F16;                                      // a scalar level in fp16
  EXPECT(list_sdf_query_fwd(&e, NULL) == LIST_ERR_UNSUPPORTED);
  e = a; e.map_size = 1;
  EXPECT(list_sdf_query_fwd(&e, NULL) == LIST_ERR_SHAPE);
  e = a; e.map_size = 46341; e.img_C = 8;                                    // map_size^2 * C just over 2^31
  EXPECT(list_sdf_query_fwd(&e, NULL) == LIST_ERR_SHAPE);
  e = a; e.clamp_hi = -1.f;
  EXPECT(list_sdf_query_fwd(&e, NULL) == LIST_ERR_ARG);
  e = a; e.F = 3611;
  EXPECT(list_sdf_query_fwd(&e, NULL) == LIST_ERR_SHAPE);
  e = a; e.H1 = 500;
  EXPECT(list_sdf_query_fwd(&e, NULL) == LIST_ERR_UNSUPPORTED);
  e = a; e.precision = 9;
  EXPECT(list_sdf_query_fwd(&e, NULL) == LIST_ERR_ARG);
  e = a; e.no_activations = 2;
  EXPECT(list_sdf_query_fwd(&e, NULL) == LIST_ERR_ARG);
  e = a; e.no_fused_fc0 = -3;
  EXPECT(list_sdf_query_fwd(&e, NULL) == LIST_ERR_ARG);
  e = a; e.workspace = (char*)e.workspace + 8;
  EXPECT(list_sdf_query_fwd(&e, NULL) == LIST_ERR_SHAPE);
  e = a; e.workspace_bytes = 1 << 20;
  EXPECT(list_sdf_query_fwd(&e, NULL) == LIST_ERR_WORKSPACE);
  e = a; e.sdf = NULL;
  EXPECT(list_sdf_query_fwd(&e, NULL) == LIST_ERR_ARG);
  e = a; e.percep_proj = fake(0xB000000); e.percep_feat = (const float*)fake(0xC000000);
  EXPECT(list_sdf_query_fwd(&e, NULL) == LIST_ERR_ARG);
  e = a; e.percep_proj = fake(0xB000000); e.img_dtype = LIST_MAP_F32;        // fp16 operands with an fp32 map
  EXPECT(list_sdf_query_fwd(&e, NULL) == LIST_ERR_UNSUPPORTED);

  // ---- projected encoder levels (ABI 8): the flag's contract, before any launch
  e = a; e.img_proj = 1; e.img_kept_C = 128; e.no_activations = 0;           // a forward a backward may follow
  EXPECT(list_sdf_query_fwd(&e, NULL) == LIST_ERR_ARG);
  e = a; e.img_proj = 2;
  EXPECT(list_sdf_query_fwd(&e, NULL) == LIST_ERR_ARG);
  e = a; e.img_kept_C = 128;                                                  // kept channels without the flag
  EXPECT(list_sdf_query_fwd(&e, NULL) == LIST_ERR_ARG);
  e = a; e.img_proj = 1; e.img_kept_C = 100;
  EXPECT(list_sdf_query_fwd(&e, NULL) == LIST_ERR_UNSUPPORTED);
  e = a; e.img_proj = 1; e.img_kept_C = 1024;                                 // nothing left to project
  EXPECT(list_sdf_query_fwd(&e, NULL) == LIST_ERR_UNSUPPORTED);
  e = a; e.img_proj = 1; e.img_kept_C = 128; e.percep_proj = fake(0xB000000);
  EXPECT(list_sdf_query_fwd(&e, NULL) == LIST_ERR_ARG);
  e = a; e.img_proj = 1; e.img_kept_C = 128; e.img_dtype = LIST_MAP_F32;      // fp16 operands with an fp32 map
  EXPECT(list_sdf_query_fwd(&e, NULL) == LIST_ERR_UNSUPPORTED);
  {
    ListMap2D lv[LIST_N_IMG_LEVELS];
    const int cs[5] = {64, 64, 128, 256, 512}, rs[5] = {224, 112, 56, 28, 14};
    for (int i = 0; i < 5; ++i) {
      lv[i].data = (const float*)fake(0x60000000); lv[i].C = cs[i]; lv[i].H = lv[i].W = rs[i];
      lv[i].sb = (int64_t)cs[i] * rs[i] * rs[i]; lv[i].sc = (int64_t)rs[i] * rs[i]; lv[i].sh = rs[i]; lv[i].sw = 1;
    }
    EXPECT(list_img_proj_map_bytes(lv, 8, 137, 2, 512, LIST_PREC_FP16) == (size_t)8 * 137 * 137 * 640 * 2);
    EXPECT(list_img_proj_map_bytes(lv, 8, 137, 5, 512, LIST_PREC_FP16) == 0);
    EXPECT(list_img_proj_map_bytes(lv, 8, 137, -1, 512, LIST_PREC_FP16) == 0);
    EXPECT(list_img_proj_map_bytes(lv, 0, 137, 2, 512, LIST_PREC_FP16) == 0);
    EXPECT(list_img_proj_map_bytes(lv, 8, 137, 2, 500, LIST_PREC_FP16) == 0);
    EXPECT(list_img_proj_scratch_bytes(lv, 8, 2, 512, 7) == 0);
    const size_t sb = list_img_proj_scratch_bytes(lv, 8, 2, 512, LIST_PREC_BF16X3);
    EXPECT(sb > 0);
    const int32_t vc[LIST_N_VOX_LEVELS] = {1, 16, 32, 64, 128, 128};
    // refused before any launch: NULL buffers, a short output, a short scratch, a misaligned scratch
    EXPECT(list_prep_img_proj(lv, 8, 137, 2, vc, fake(0x9000000), 512, 256, 256, LIST_PREC_BF16X3, NULL, 0, fake(0x70000000), sb, NULL) == LIST_ERR_ARG);
    EXPECT(list_prep_img_proj(lv, 8, 137, 2, vc, fake(0x9000000), 512, 256, 256, LIST_PREC_BF16X3, fake(0x80000000), 1000, fake(0x70000000), sb, NULL) == LIST_ERR_WORKSPACE);
    EXPECT(list_prep_img_proj(lv, 8, 137, 2, vc, fake(0x9000000), 512, 256, 256, LIST_PREC_BF16X3, fake(0x80000000), (size_t)1 << 40, fake(0x70000000), sb - 1, NULL) == LIST_ERR_WORKSPACE);
    EXPECT(list_prep_img_proj(lv, 8, 137, 2, vc, fake(0x9000000), 512, 256, 256, LIST_PREC_BF16X3, fake(0x80000000), (size_t)1 << 40, fake(0x70000008), sb, NULL) == LIST_ERR_SHAPE);
    EXPECT(list_prep_img_proj(lv, 8, 400, 2, vc, fake(0x9000000), 512, 256, 256, LIST_PREC_BF16X3, fake(0x80000000), (size_t)1 << 40, fake(0x70000000), sb, NULL) == LIST_ERR_SHAPE);
    lv[4].C = 500;
    EXPECT(list_prep_img_proj(lv, 8, 137, 2, vc, fake(0x9000000), 512, 256, 256, LIST_PREC_BF16X3, fake(0x80000000), (size_t)1 << 40, fake(0x70000000), sb, NULL) == LIST_ERR_UNSUPPORTED);
  }

  // ---- the plan: what the library would dispatch, from the same validation (no launch)
  ListQueryPlan pl;
  EXPECT(list_query_plan(&a, NULL) == LIST_ERR_ARG);
  EXPECT(list_query_plan(NULL, &pl) == LIST_ERR_ARG);
  EXPECT(list_query_plan(&a, &pl) == LIST_OK && pl.chunks == 1 && pl.rows_per_chunk == 160000 && pl.fused_tail == 1 &&
         pl.fc0_k == 3648 && pl.box_levels == ((1 << 4) | (1 << 5)));
  EXPECT(pl.fused_fc0 == 1);                                                  // inference forward, fp16 operands and maps
  EXPECT(pl.img_proj == 0);
  e = a; e.no_fused_fc0 = 1;
  EXPECT(list_query_plan(&e, &pl) == LIST_OK && pl.fused_fc0 == 0 && pl.fused_tail == 1);
  e = a; e.img_proj = 1; e.img_kept_C = 128;                                  // fc_0 without the projected levels' 896 columns
  EXPECT(list_query_plan(&e, &pl) == LIST_OK && pl.img_proj == 1 && pl.fc0_k == 3648 - 896 && pl.fused_fc0 == 1);
  e.precision = LIST_PREC_BF16X3; e.img_dtype = LIST_MAP_F32;                 // the bf16 formats: 2-D gather + row-vector epilogue
  EXPECT(list_query_plan(&e, &pl) == LIST_OK && pl.img_proj == 1 && pl.fc0_k == 3648 - 896 && pl.fused_fc0 == 0);
  e = a; e.no_activations = 0;
  EXPECT(list_query_plan(&e, &pl) == LIST_OK && pl.fused_tail == 0 && pl.fused_fc0 == 0);
  e = a; e.precision = LIST_PREC_BF16X3;
  EXPECT(list_query_plan(&e, &pl) == LIST_OK && pl.fused_tail == 0 && pl.box_levels == 0);
  ListQueryArgs g = good_query(1, 256 * 256 * 256, fake(0x40000000), ws_grid);
  EXPECT(list_query_plan(&g, &pl) == LIST_OK && pl.chunks == 64 && pl.rows_per_chunk == 262144);
  g.workspace_bytes = ws_grid / 3;                                            // a third of the workspace: more, smaller chunks
  EXPECT(list_query_plan(&g, &pl) == LIST_OK && pl.chunks > 64 * 3 - 3 && pl.rows_per_chunk % 256 == 0 &&
         (int64_t)pl.chunks * pl.rows_per_chunk >= (int64_t)256 * 256 * 256);
  g = good_query(46341, 46341, fake(0x40000000), ws_grid);                    // B * N just over 2^31 points
  EXPECT(list_query_plan(&g, &pl) == LIST_OK && pl.rows_per_chunk == 262144 && pl.chunks == (int32_t)(((int64_t)46341 * 46341 + 262143) / 262144));
  e = a; e.B = 0;
  EXPECT(list_query_plan(&e, &pl) == LIST_OK && pl.chunks == 0);

  // ---- backward: refused before anything is enqueued
  ListQueryGradArgs ga;
  memset(&ga, 0, sizeof(ga));
  EXPECT(list_sdf_query_bwd(NULL, NULL) == LIST_ERR_ARG);
  EXPECT(list_sdf_query_bwd(&ga, NULL) == LIST_ERR_ARG);
  ga.fwd = &a; ga.grad_sdf = (const float*)fake(0xD000000); ga.packed_mlp_bwd = fake(0xE000000);
  ga.workspace = fake(0x50000000); ga.workspace_bytes = (size_t)1 << 40;
  EXPECT(list_sdf_query_bwd(&ga, NULL) == LIST_ERR_ARG);                      // an inference forward keeps no activations
  e = a; e.no_activations = 0; e.B = 14; ga.fwd = &e;                          // 280 000 points: more than one chunk
  EXPECT(list_sdf_query_bwd(&ga, NULL) == LIST_ERR_UNSUPPORTED);
  e = a; e.no_activations = 0; e.workspace_bytes = ws160k / 2; ga.fwd = &e;
  EXPECT(list_sdf_query_bwd(&ga, NULL) == LIST_ERR_WORKSPACE);
  e = a; e.no_activations = 0; ga.fwd = &e; ga.workspace_bytes = 1 << 20;
  EXPECT(list_sdf_query_bwd(&ga, NULL) == LIST_ERR_WORKSPACE);
  ga.workspace_bytes = (size_t)1 << 40; ga.vox_adjoint = 5;
  EXPECT(list_sdf_query_bwd(&ga, NULL) == LIST_ERR_ARG);
  ga.vox_adjoint = 0; ga.grad_img_map_dtype = 3;
  EXPECT(list_sdf_query_bwd(&ga, NULL) == LIST_ERR_ARG);

  // ---- the map hand-offs and the weight repack
  ListMap2D im[LIST_N_IMG_LEVELS];
  memset(im, 0, sizeof(im));
  const int ic[5] = {64, 64, 128, 256, 512}, ir[5] = {224, 112, 56, 28, 14};
  for (int i = 0; i < 5; ++i) {
    im[i].data = (const float*)fake(0x1000000 * (i + 1)); im[i].C = ic[i]; im[i].H = im[i].W = ir[i];
    im[i].sb = (int64_t)ic[i] * ir[i] * ir[i]; im[i].sc = (int64_t)ir[i] * ir[i]; im[i].sh = ir[i]; im[i].sw = 1;
  }
  EXPECT(list_img_map_bytes(im, 8, 137, LIST_MAP_F16) == (size_t)8 * 137 * 137 * 1024 * 2);
  EXPECT(list_img_map_bytes(im, 0, 137, LIST_MAP_F16) == 0 && list_img_map_bytes(NULL, 8, 137, 0) == 0);
  EXPECT(list_prep_img_maps(NULL, 8, 137, 0, fake(0), 1, NULL) == LIST_ERR_ARG);
  EXPECT(list_prep_img_maps(im, 8, 1, 0, fake(0), (size_t)1 << 40, NULL) == LIST_ERR_SHAPE);
  EXPECT(list_prep_img_maps(im, 8, 321, 0, fake(0), (size_t)1 << 40, NULL) == LIST_ERR_SHAPE);
  EXPECT(list_prep_img_maps(im, 8, 137, 5, fake(0), (size_t)1 << 40, NULL) == LIST_ERR_ARG);
  EXPECT(list_prep_img_maps(im, 8, 137, LIST_MAP_F16, fake(0), 1000, NULL) == LIST_ERR_WORKSPACE);
  EXPECT(list_prep_img_maps(im, 8, 137, LIST_MAP_F16, fake(4), (size_t)1 << 40, NULL) == LIST_ERR_SHAPE);
  im[4].C = 510;
  EXPECT(list_prep_img_maps(im, 8, 137, LIST_MAP_F16, fake(0), (size_t)1 << 40, NULL) == LIST_ERR_UNSUPPORTED);
  im[4].C = 512; im[2].H = 0;
  EXPECT(list_prep_img_maps(im, 8, 137, LIST_MAP_F16, fake(0), (size_t)1 << 40, NULL) == LIST_ERR_SHAPE);

  ListMap3D vm[LIST_N_VOX_LEVELS];
  ListVoxLevel lv[LIST_N_VOX_LEVELS];
  memset(vm, 0, sizeof(vm));
  const int vc[6] = {1, 16, 32, 64, 128, 128}, vr[6] = {128, 128, 64, 32, 16, 8};
  for (int l = 0; l < 6; ++l) {
    vm[l].data = fake(0x10000000 * (uintptr_t)(l + 1)); vm[l].C = vc[l]; vm[l].D = vm[l].H = vm[l].W = vr[l];
    const int64_t v = (int64_t)vr[l] * vr[l] * vr[l];
    vm[l].sb = vc[l] * v; vm[l].sc = v; vm[l].sd = (int64_t)vr[l] * vr[l]; vm[l].sh = vr[l]; vm[l].sw = 1; vm[l].dtype = LIST_MAP_F32;
  }
  const size_t pack = list_vox_pack_bytes(vm, 8, LIST_MAP_F16);
  EXPECT(pack >= (size_t)8 * 2 * ((size_t)16 * 2097152 + 32 * 262144 + 64 * 32768 + 128 * 4096 + 128 * 512));
  EXPECT(list_vox_pack_bytes(vm, 0, LIST_MAP_F16) == 0 && list_vox_pack_bytes(vm, 8, 9) == 0);
  EXPECT(list_prep_vox_maps(NULL, 8, 0, fake(0), pack, lv, NULL) == LIST_ERR_ARG);
  EXPECT(list_prep_vox_maps(vm, 65536, 0, fake(0), (size_t)1 << 44, lv, NULL) == LIST_ERR_SHAPE);
  EXPECT(list_prep_vox_maps(vm, 8, LIST_MAP_F16, fake(0), pack - 1, lv, NULL) == LIST_ERR_WORKSPACE);
  EXPECT(list_prep_vox_maps(vm, 8, LIST_MAP_F16, fake(8), pack, lv, NULL) == LIST_ERR_SHAPE);
  EXPECT(list_prep_vox_maps(vm, 8, LIST_MAP_F16, NULL, 0, lv, NULL) == LIST_ERR_WORKSPACE);
  vm[3].dtype = 4;
  EXPECT(list_prep_vox_maps(vm, 8, LIST_MAP_F16, fake(0), pack, lv, NULL) == LIST_ERR_ARG);

  ListMlpWeights w;
  memset(&w, 0, sizeof(w));
  EXPECT(list_packed_mlp_bytes(NULL) == 0 && list_packed_mlp_bytes(&w) == 0);
  EXPECT(list_prep_mlp_weights(NULL, fake(0), 1, NULL) == LIST_ERR_ARG);
  w.w0 = w.b0 = w.w1 = w.b1 = w.w2 = w.b2 = w.w3 = w.b3 = (const float*)fake(0x6000000);
  for (int l = 0; l < 6; ++l) w.vox_C[l] = vc[l];
  w.img_C = 1024; w.F = 3610; w.H1 = 512; w.H2 = 256; w.H3 = 256; w.precision = LIST_PREC_FP16;
  const size_t pk = list_packed_mlp_bytes(&w);
  EXPECT(pk > (size_t)512 * 3648 * 2);
  EXPECT(list_prep_mlp_weights(&w, fake(0), pk - 1, NULL) == LIST_ERR_WORKSPACE);
  EXPECT(list_prep_mlp_weights(&w, fake(2), pk, NULL) == LIST_ERR_SHAPE);
  w.F = 3609;
  EXPECT(list_prep_mlp_weights(&w, fake(0), pk, NULL) == LIST_ERR_SHAPE);
  w.F = 3610; w.H3 = 128;
  EXPECT(list_prep_mlp_weights(&w, fake(0), pk, NULL) == LIST_ERR_UNSUPPORTED);
  w.H3 = 256; w.precision = -1;
  EXPECT(list_prep_mlp_weights(&w, fake(0), pk, NULL) == LIST_ERR_ARG);
  w.precision = LIST_PREC_FP16; w.vox_C[2] = 30;
  EXPECT(list_prep_mlp_weights(&w, fake(0), pk, NULL) == LIST_ERR_UNSUPPORTED);

  ListPoolArgs pa;
  memset(&pa, 0, sizeof(pa));
  EXPECT(list_percep_pool_fwd(NULL, NULL) == LIST_ERR_ARG);
  EXPECT(list_percep_pool_fwd(&pa, NULL) == LIST_ERR_ARG);

  if (g_fail) { fprintf(stderr, "capi_host_asan: %d expectation(s) failed\n", g_fail); return 1; }
  printf("CAPI_HOST_SANITIZERS_OK\n");
  return 0;
}
