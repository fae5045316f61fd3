// C ABI (include/list_hip.h): argument validation, workspace carving, launch sequencing.
// No allocation, no synchronisation, no global mutable state (thread-local error text only).
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "list_common.h"

using namespace list;

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int hip_fail(hipError_t e, const char* what) {
  return fail(LIST_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
}

constexpr int64_t kMaxChunkRows = 262144;     // bounds the workspace (~4.6 GB) for huge queries

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

bool vox_in_place(const ListMap3D& m, ListVoxLevel* lv, int32_t map_dtype) {
  const int64_t C = m.C, W = m.W, H = m.H;
  lv->C = m.C; lv->D = m.D; lv->H = m.H; lv->W = m.W; lv->dtype = LIST_MAP_F32; lv->reserved_ = 0;
  if (m.dtype == LIST_MAP_F16) {
    // a half-precision producer: its channels-last levels are the fp16 maps themselves; scalar levels and
    // requests for fp32 maps go through the converting copy
    if (map_dtype != LIST_MAP_F16 || m.C == 1 || (m.C % 8) != 0) return false;
    if (m.sc == 1 && m.sw == C && m.sh == W * C && m.sd == H * W * C && aligned16(m.data) && (m.sb % 8) == 0) {
      lv->data = m.data; lv->image_stride = m.sb; lv->dtype = LIST_MAP_F16; return true;
    }
    return false;
  }
  if (m.C == 1) {
    if (m.sw == 1 && m.sh == W && m.sd == H * W) { lv->data = m.data; lv->image_stride = m.sb; return true; }
    return false;
  }
  // fp16 maps asked for and the level is one the matrix-core gather takes (128 channels, at most 31 voxels a side:
  // k_gather_vox_box wants fp16 voxels): a channels-last fp32 level is then CONVERTED (1 - 8 MB per batch, a few
  // microseconds) rather than used in place -- in place it fell back to the scalar shared-tap kernel on 4-byte taps
  // (round 4, the module path on channels-last encoders: levels 4 + 5 0.27 -> 0.20 ms)
  if (map_dtype == LIST_MAP_F16 && m.C == 128 && m.D <= 31 && m.H <= 31 && m.W <= 31) return false;
  if (m.sc == 1 && m.sw == C && m.sh == W * C && m.sd == H * W * C && aligned16(m.data) &&
      (m.sb % 4) == 0) {
    lv->data = m.data; lv->image_stride = m.sb; return true;
  }
  return false;
}

int check_query_common(const ListQueryArgs* a, FeatLayout* L) {
  if (!a) return fail(LIST_ERR_ARG, "args is NULL");
  if (a->B <= 0 || a->N <= 0) return fail(LIST_ERR_SHAPE, "B=%d N=%d must be positive", a->B, a->N);
  if (!a->query || !a->workspace) return fail(LIST_ERR_ARG, "query/workspace is NULL");
  for (int i = 0; i < 3; ++i)
    if (a->perm[i] < 0 || a->perm[i] > 2) return fail(LIST_ERR_ARG, "perm[%d]=%d", i, a->perm[i]);
  int32_t vc[LIST_N_VOX_LEVELS];
  for (int l = 0; l < LIST_N_VOX_LEVELS; ++l) {
    const ListVoxLevel& v = a->vox[l];
    if (!v.data || v.D < 1 || v.H < 1 || v.W < 1)
      return fail(LIST_ERR_SHAPE, "voxel level %d: bad descriptor", l);
    if ((int64_t)v.D * v.H * v.W * v.C >= (int64_t)1 << 31)
      return fail(LIST_ERR_SHAPE, "voxel level %d: image larger than 2^31 elements", l);
    if (v.dtype != LIST_MAP_F32 && v.dtype != LIST_MAP_F16)
      return fail(LIST_ERR_ARG, "voxel level %d: dtype=%d", l, v.dtype);
    if (v.C != 1 && (!aligned16(v.data) || (v.image_stride % (v.dtype == LIST_MAP_F16 ? 8 : 4)) != 0))
      return fail(LIST_ERR_SHAPE, "voxel level %d: data must be 16-byte aligned", l);
    if (v.dtype == LIST_MAP_F16 && (v.C == 1 || v.C % 8 != 0))
      return fail(LIST_ERR_UNSUPPORTED, "voxel level %d: fp16 maps need C %% 8 == 0 (C=%d)", l, v.C);
    if (v.C != 1 && (v.C > 256 || (v.C & (v.C - 1)) != 0 || v.C < 4))
      return fail(LIST_ERR_UNSUPPORTED, "voxel level %d: C=%d (need 1 or a power of two in 4..256)", l, v.C);
    vc[l] = v.C;
  }
  if (!make_layout(vc, a->img_C, L)) return fail(LIST_ERR_UNSUPPORTED, "unsupported channel counts");
  if (a->percep_feat == nullptr) {
    if (!a->img_map || !a->trans_mat) return fail(LIST_ERR_ARG, "img_map/trans_mat is NULL");
    if (!aligned16(a->img_map)) return fail(LIST_ERR_SHAPE, "img_map must be 16-byte aligned");
    if (a->img_dtype != LIST_MAP_F32 && a->img_dtype != LIST_MAP_F16)
      return fail(LIST_ERR_ARG, "img_dtype=%d", a->img_dtype);
    if (a->img_dtype == LIST_MAP_F16 && a->img_C % 8)
      return fail(LIST_ERR_UNSUPPORTED, "fp16 image map needs img_C %% 8 == 0");
    if (a->map_size < 2) return fail(LIST_ERR_SHAPE, "map_size=%d", a->map_size);
    if (!(a->clamp_hi >= 0.f)) return fail(LIST_ERR_ARG, "clamp_hi=%g must be >= 0", (double)a->clamp_hi);
    if ((int64_t)a->map_size * a->map_size * a->img_C >= (int64_t)1 << 31)
      return fail(LIST_ERR_SHAPE, "image map larger than 2^31 elements");
  }
  if (a->img_proj != 0 && a->img_proj != 1)
    return fail(LIST_ERR_ARG, "img_proj=%d (0 or 1; zero-initialise ListQueryArgs, list_abi_version() = %d)", a->img_proj,
                LIST_ABI_VERSION);
  if (a->img_proj) {
    // list_prep_img_proj's map: img_kept_C sampled channels | H1 projected ones; the row vector lives in the lo plane of
    // X (fp16 operands: unused otherwise) or behind the kept columns of the row's perceptual block (split formats)
    if (a->percep_feat || a->percep_proj) return fail(LIST_ERR_ARG, "img_proj excludes percep_feat / percep_proj");
    if (!a->no_activations) return fail(LIST_ERR_ARG, "img_proj is an inference path: set no_activations = 1");
    const int xbytes = a->precision == LIST_PREC_FP16 ? 2 : 4;
    const int ktile = a->precision == LIST_PREC_FP16 ? 64 : 32;            // columns per 128-byte K-tile of X
    if (a->img_kept_C < 0 || a->img_kept_C >= a->img_C || a->img_kept_C % 64 || a->img_C % ktile || a->H1 % 8)
      return fail(LIST_ERR_UNSUPPORTED, "img_proj needs 0 <= img_kept_C < img_C, img_kept_C %% 64 == 0, H1 %% 8 == 0 "
                                        "(img_kept_C=%d, img_C=%d, H1=%d)", a->img_kept_C, a->img_C, a->H1);
    if (xbytes == 4 && (int64_t)(a->img_kept_C + a->H1) * 4 > (int64_t)a->img_C * 4)
      return fail(LIST_ERR_UNSUPPORTED, "img_proj: img_kept_C + H1 must fit the perceptual block (%d + %d > %d)",
                  a->img_kept_C, a->H1, a->img_C);
    if ((int64_t)a->map_size * a->map_size * (a->img_kept_C + a->H1) >= (int64_t)1 << 31)
      return fail(LIST_ERR_SHAPE, "projected map larger than 2^31 elements");
    if ((a->precision == LIST_PREC_FP16) != (a->img_dtype == LIST_MAP_F16))
      return fail(LIST_ERR_UNSUPPORTED, "img_proj: fp16 operands pair with fp16 maps, the bf16 formats with fp32 maps");
  } else if (a->img_kept_C != 0) {
    return fail(LIST_ERR_ARG, "img_kept_C=%d without img_proj", a->img_kept_C);
  }
  if (a->percep_proj) {
    // the projected sample (H1 floats) lives in the perceptual block of the point's X row; fc_0 starts behind it
    if (a->percep_feat) return fail(LIST_ERR_ARG, "percep_proj and percep_feat exclude each other");
    const int xbytes = a->precision == LIST_PREC_FP16 ? 2 : 4;            // per column of X (hi + lo in the split formats)
    if (!aligned16(a->percep_proj) || a->img_C % 64 || a->H1 % 8 || (int64_t)a->H1 * 4 > (int64_t)a->img_C * xbytes)
      return fail(LIST_ERR_UNSUPPORTED, "percep_proj needs img_C %% 64 == 0, H1 %% 8 == 0 and H1 * 4 <= img_C * %d bytes", xbytes);
    if ((int64_t)a->map_size * a->map_size * a->H1 >= (int64_t)1 << 31)
      return fail(LIST_ERR_SHAPE, "projected map larger than 2^31 elements");
    if ((a->precision == LIST_PREC_FP16) != (a->img_dtype == LIST_MAP_F16))
      return fail(LIST_ERR_UNSUPPORTED, "percep_proj: fp16 operands pair with fp16 maps, the bf16 formats with fp32 maps");
  }
  return LIST_OK;
}

GatherParams make_gather(const ListQueryArgs* a, const FeatLayout& L, const Workspace& ws,
                         int64_t p_begin, int n_valid, int rows) {
  GatherParams g;
  g.query = a->query; g.q_sb = a->q_sb; g.q_sn = a->q_sn; g.q_sc = a->q_sc;
  g.perm0 = a->perm[0]; g.perm1 = a->perm[1]; g.perm2 = a->perm[2]; g.scale = a->scale;
  g.N = a->N; g.p_begin = p_begin; g.n_valid = n_valid; g.rows = rows;
  g.x_hi = (unsigned short*)((char*)a->workspace + ws.x_hi);
  g.x_lo = (unsigned short*)((char*)a->workspace + ws.x_lo);
  g.Kp = L.Kp;
  g.fmt = a->precision == LIST_PREC_FP16 ? FMT_FP16 : FMT_BF16_SPLIT;
  g.order = nullptr; g.order_img = nullptr; g.row_of = nullptr;
  // where the projected perceptual sample (H1 floats per row) goes; fc_0's epilogue reads it from there
  g.rowvec = nullptr; g.rv_stride = 0;
  const int xb = a->precision == LIST_PREC_FP16 ? 2 : 4;
  if (a->percep_proj) {                      // the head of the row's perceptual block
    g.rowvec = (float*)g.x_hi; g.rv_stride = (int64_t)L.Kp * xb / 4;
  } else if (a->img_proj) {
    if (xb == 2) { g.rowvec = (float*)g.x_lo; g.rv_stride = a->H1; }                 // the lo plane: rows * Kp * 2 >= rows * H1 * 4
    else { g.rowvec = (float*)g.x_hi + a->img_kept_C; g.rv_stride = (int64_t)L.Kp; }  // behind the kept columns (4 B per column)
  }
  return g;
}

// rows the given workspace can hold (multiple of kRowTile, capped at the request), or 0
int64_t chunk_rows_for(size_t bytes, int64_t P, int Kp, int H1, int H2) {
  // (clamp first: P comes from the caller, and P + 255 must not overflow -- found by the UBSan host build)
  int64_t want = P >= kMaxChunkRows ? kMaxChunkRows : (P + kRowTile - 1) / kRowTile * kRowTile;
  if (workspace_layout(want, Kp, H1, H2).total <= bytes) return want;
  if (bytes < workspace_fixed_bytes()) return 0;
  int64_t fit = (int64_t)((bytes - workspace_fixed_bytes()) / workspace_row_bytes(Kp, H1, H2) + kRowTile)
                / kRowTile * kRowTile;
  if (fit > want) fit = want;
  while (fit >= kRowTile && workspace_layout(fit, Kp, H1, H2).total > bytes) fit -= kRowTile;
  return fit >= kRowTile ? fit : 0;
}

// fc_0 with the perceptual block of its A operand produced on chip (fused_fc0_kernels.hip): fp16 operands and maps,
// the per-point 2-D sample (no pre-pooled features, no projected map), H1 = 512.  LIST_FUSED_FC0 (environment, read
// once): "0" keeps the unfused path (2-D gather kernel + k_gemm_nt_pp), "x" runs the 128 x 512 tile with every K-tile
// from X (tile-shape diagnostic: the 2-D gather still runs).
int fused_fc0_mode() {
  static const int mode = [] {
    const char* e = getenv("LIST_FUSED_FC0");
#ifdef LIST_FUSED_FC0_DEFAULT_OFF
    if (!e || !e[0]) return 0;
#else
    if (!e || !e[0]) return 1;
#endif
    if (e[0] == '0' && e[1] == 0) return 0;
    if (e[0] == 'x') return 2;
    if (e[0] == '3' && e[1] == 0) return 3;
    return 1;
  }();
  return mode;
}
bool takes_fused_fc0_any(const ListQueryArgs* a, const FeatLayout& L) {
  if (fused_fc0_mode() == 0) return false;
  if (a->percep_feat || a->percep_proj) return false;
  // list_prep_img_proj's map: the kept levels are produced on chip, the projected channels sampled in the epilogue
  // (k_fc0_fused<0, true>: fp16 operands only)
  if (a->img_proj && (a->precision != LIST_PREC_FP16 || fused_fc0_mode() == 2)) return false;
  // fp16 operands with fp16 maps, the bf16 formats with fp32 maps (the pairs the standard path takes too)
  if ((a->precision == LIST_PREC_FP16) != (a->img_dtype == LIST_MAP_F16)) return false;
  // bf16x3 keeps the unfused path: its packed weight is twice as long (hi + lo), and a 128-row tile streams ALL of it
  // (7.3 MB per tile, 9.3 GB per 160 k points) -- measured (round 4): fc_0 1.15 -> 2.15 ms, step 3.58 -> 4.20 ms.
  // LIST_FUSED_FC0=3 forces it (A/B runs, tests).
  if (a->precision == LIST_PREC_BF16X3 && fused_fc0_mode() != 3) return false;
  // inference forwards only: list_sdf_query_bwd reads the WHOLE feature matrix (d fc_0.weight = dZ1^T . X, the perceptual
  // columns included), so a forward that a backward may follow materialises it
  if (!a->no_activations || a->no_fused_fc0) return false;
  return a->H1 == 512 && a->img_C > 0 && a->img_C % 64 == 0 && L.img_off == 0 && L.Kp % 64 == 0;
}
int fused_produced_tiles(const ListQueryArgs* a) {
  const int channels = a->img_proj ? a->img_kept_C : a->img_C;
  return fused_fc0_mode() == 2 ? 0 : channels / (a->precision == LIST_PREC_FP16 ? 64 : 32);
}
// true: the 2-D gather kernel is NOT launched (its columns are produced inside fc_0)
bool takes_fused_fc0(const ListQueryArgs* a, const FeatLayout& L) {
  return takes_fused_fc0_any(a, L) && fused_fc0_mode() != 2;
}

// inference forwards in fp16: fc_1, fc_2 and fc_out as ONE launch (gemm_kernels.hip, k_mlp_tail_f16).  The one
// predicate behind the dispatch in list_sdf_query_fwd and behind list_query_plan (what a caller's accounting reads).
bool takes_fused_tail(const ListQueryArgs* a) {
#ifdef LIST_NO_FUSED_TAIL
  (void)a;
  return false;
#else
  return a->no_activations && a->precision == LIST_PREC_FP16 && a->H2 == 256 && a->H3 == 256 && a->H1 % 64 == 0;
#endif
}

}  // namespace

extern "C" {

const char* list_last_error(void) { return g_err; }
int list_abi_version(void) { return LIST_ABI_VERSION; }

// ------------------------------------------------------------------------------------------ 2-D maps
static bool dtype_ok(int32_t d) { return d == LIST_MAP_F32 || d == LIST_MAP_F16; }
static size_t elem_bytes(int32_t d) { return d == LIST_MAP_F16 ? 2 : 4; }

size_t list_img_map_bytes(const ListMap2D maps[LIST_N_IMG_LEVELS], int32_t B, int32_t map_size,
                          int32_t map_dtype) {
  if (!maps || B <= 0 || map_size <= 0 || !dtype_ok(map_dtype)) return 0;
  size_t Ct = 0;
  for (int i = 0; i < LIST_N_IMG_LEVELS; ++i) Ct += (size_t)maps[i].C;
  return (size_t)B * map_size * map_size * Ct * elem_bytes(map_dtype);
}

int list_prep_img_maps(const ListMap2D maps[LIST_N_IMG_LEVELS], int32_t B, int32_t map_size,
                       int32_t map_dtype, void* out, size_t out_bytes, void* stream) {
  if (!maps || !out) return fail(LIST_ERR_ARG, "maps/out is NULL");
  if (!dtype_ok(map_dtype)) return fail(LIST_ERR_ARG, "map_dtype=%d", map_dtype);
  if (B <= 0 || map_size < 2 || map_size > 320)
    return fail(LIST_ERR_SHAPE, "B=%d map_size=%d (need 2..320)", B, map_size);
  int Ct = 0;
  for (int i = 0; i < LIST_N_IMG_LEVELS; ++i) {
    const ListMap2D& m = maps[i];
    if (!m.data || m.C < 1 || m.H < 1 || m.W < 1)
      return fail(LIST_ERR_SHAPE, "image level %d: bad descriptor", i);
    Ct += m.C;
  }
  const int align = map_dtype == LIST_MAP_F16 ? 8 : 4;
  if (Ct % align) return fail(LIST_ERR_UNSUPPORTED, "total image channels %d not a multiple of %d", Ct, align);
  if (!aligned16(out)) return fail(LIST_ERR_SHAPE, "out must be 16-byte aligned");
  if (out_bytes < list_img_map_bytes(maps, B, map_size, map_dtype))
    return fail(LIST_ERR_WORKSPACE, "out buffer too small: %zu < %zu", out_bytes,
                list_img_map_bytes(maps, B, map_size, map_dtype));
  if ((int64_t)B * map_size > 2147483647LL) return fail(LIST_ERR_SHAPE, "grid too large");
  hipError_t e = launch_prep_img(maps, B, map_size, Ct, map_dtype == LIST_MAP_F16, out, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "prep_img launch");
  return LIST_OK;
}

// ------------------------------------------------------------------------------------------ 3-D maps
static bool level_as_f16(const ListMap3D& m, int32_t map_dtype) {
  return map_dtype == LIST_MAP_F16 && m.C != 1 && (m.C % 8) == 0;
}

size_t list_vox_pack_bytes(const ListMap3D maps[LIST_N_VOX_LEVELS], int32_t B, int32_t map_dtype) {
  if (!maps || B <= 0 || !dtype_ok(map_dtype)) return 0;
  size_t total = 0;
  for (int l = 0; l < LIST_N_VOX_LEVELS; ++l) {
    ListVoxLevel lv;
    if (!dtype_ok(maps[l].dtype)) return 0;
    if (vox_in_place(maps[l], &lv, map_dtype)) continue;
    total += align_up((size_t)B * maps[l].C * maps[l].D * maps[l].H * maps[l].W *
                      (level_as_f16(maps[l], map_dtype) ? 2 : 4), 256);
  }
  return total;
}

int list_prep_vox_maps(const ListMap3D maps[LIST_N_VOX_LEVELS], int32_t B, int32_t map_dtype,
                       void* pack, size_t pack_bytes, ListVoxLevel levels_out[LIST_N_VOX_LEVELS],
                       void* stream) {
  if (!maps || !levels_out) return fail(LIST_ERR_ARG, "maps/levels_out is NULL");
  if (!dtype_ok(map_dtype)) return fail(LIST_ERR_ARG, "map_dtype=%d", map_dtype);
  if (B <= 0 || B > 65535) return fail(LIST_ERR_SHAPE, "B=%d", B);
  for (int l = 0; l < LIST_N_VOX_LEVELS; ++l)
    if (!dtype_ok(maps[l].dtype)) return fail(LIST_ERR_ARG, "voxel level %d: dtype=%d", l, maps[l].dtype);
  const size_t need = list_vox_pack_bytes(maps, B, map_dtype);
  if (need > 0 && (!pack || pack_bytes < need))
    return fail(LIST_ERR_WORKSPACE, "pack buffer too small: %zu < %zu", pack_bytes, need);
  if (need > 0 && !aligned16(pack)) return fail(LIST_ERR_SHAPE, "pack must be 16-byte aligned");
  size_t off = 0;
  ListMap3D fused_maps[LIST_N_VOX_LEVELS];
  void* fused_out[LIST_N_VOX_LEVELS];
  int fused_f16[LIST_N_VOX_LEVELS];
  int n_fused = 0;
  for (int l = 0; l < LIST_N_VOX_LEVELS; ++l) {
    const ListMap3D& m = maps[l];
    if (!m.data || m.C < 1 || m.D < 1 || m.H < 1 || m.W < 1)
      return fail(LIST_ERR_SHAPE, "voxel level %d: bad descriptor", l);
    if ((int64_t)m.D * m.H * m.W * m.C >= (int64_t)1 << 31)
      return fail(LIST_ERR_SHAPE, "voxel level %d: image larger than 2^31 elements", l);
    if (vox_in_place(m, &levels_out[l], map_dtype)) continue;
    const bool f16 = level_as_f16(m, map_dtype);
    void* dst = (char*)pack + off;
    // a dense channels-last fp32 level that only changes its element type (the matrix-core gather's levels, see
    // vox_in_place): one elementwise pass, no transposition
    const int64_t vol = (int64_t)m.D * m.H * m.W;
    if (f16 && m.dtype == LIST_MAP_F32 && m.sc == 1 && m.sw == m.C && m.sh == (int64_t)m.W * m.C &&
        m.sd == (int64_t)m.H * m.W * m.C && m.sb == vol * m.C && aligned16(m.data) && ((vol * m.C * B) % 4) == 0) {
      hipError_t e = launch_split((const float*)m.data, (unsigned short*)dst, nullptr, vol * m.C * B, FMT_FP16,
                                  (hipStream_t)stream);
      if (e != hipSuccess) return hip_fail(e, "fp16 conversion launch");
    } else
#ifndef LIST_TRANSPOSE_NO_FUSE
    if (transpose_tile_eligible(m, dst)) {           // launched together below
      fused_maps[n_fused] = m; fused_out[n_fused] = dst; fused_f16[n_fused] = f16 ? 1 : 0; ++n_fused;
    } else
#endif
    {
      hipError_t e = launch_transpose_vox(m, B, f16, dst, (hipStream_t)stream);
      if (e != hipSuccess) return hip_fail(e, "transpose_vox launch");
    }
    levels_out[l].data = dst;
    levels_out[l].C = m.C; levels_out[l].D = m.D; levels_out[l].H = m.H; levels_out[l].W = m.W;
    levels_out[l].dtype = f16 ? LIST_MAP_F16 : LIST_MAP_F32; levels_out[l].reserved_ = 0;
    levels_out[l].image_stride = (int64_t)m.C * m.D * m.H * m.W;
    off += align_up((size_t)B * m.C * m.D * m.H * m.W * (f16 ? 2 : 4), 256);
  }
  if (n_fused > 0) {
    hipError_t e = launch_transpose_vox_fused(fused_maps, fused_out, fused_f16, n_fused, B, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "transpose_vox launch");
  }
  return LIST_OK;
}

// ------------------------------------------------------------------------------------------ weights
static int weights_layout(const ListMlpWeights* w, FeatLayout* L) {
  if (!w) return fail(LIST_ERR_ARG, "weights is NULL");
  if (w->precision < LIST_PREC_BF16X3 || w->precision > LIST_PREC_FP16)
    return fail(LIST_ERR_ARG, "weights precision=%d", w->precision);
  if (!w->w0 || !w->b0 || !w->w1 || !w->b1 || !w->w2 || !w->b2 || !w->w3 || !w->b3)
    return fail(LIST_ERR_ARG, "a weight pointer is NULL");
  if (!make_layout(w->vox_C, w->img_C, L)) return fail(LIST_ERR_UNSUPPORTED, "unsupported channel counts");
  if (L->F != w->F) return fail(LIST_ERR_SHAPE, "F=%d but channels give %d", w->F, L->F);
  if (w->H1 % 256 || w->H2 % 256 || w->H3 != 256 || w->H1 <= 0 || w->H2 <= 0)
    return fail(LIST_ERR_UNSUPPORTED, "hidden sizes %d/%d/%d (need H1,H2 multiples of 256, H3 = 256)",
                w->H1, w->H2, w->H3);
  return LIST_OK;
}

size_t list_packed_mlp_bytes(const ListMlpWeights* w) {
  FeatLayout L;
  if (weights_layout(w, &L) != LIST_OK) return 0;
  return packed_mlp_layout(L.Kp, w->H1, w->H2, w->H3).total;
}

int list_prep_mlp_weights(const ListMlpWeights* w, void* packed, size_t packed_bytes, void* stream) {
  FeatLayout L;
  int rc = weights_layout(w, &L);
  if (rc != LIST_OK) return rc;
  const PackedMlp P = packed_mlp_layout(L.Kp, w->H1, w->H2, w->H3);
  if (!packed || packed_bytes < P.total)
    return fail(LIST_ERR_WORKSPACE, "packed buffer too small: %zu < %zu", packed_bytes, P.total);
  if (!aligned16(packed) || !aligned16(w->w1) || !aligned16(w->w2))
    return fail(LIST_ERR_SHAPE, "packed/w1/w2 must be 16-byte aligned");
  hipError_t e = launch_prep_weights(*w, L, P, (char*)packed, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "prep_weights launch");
  return LIST_OK;
}

// ------------------------------------------------------------------------------------------ query
size_t list_query_workspace_bytes(int64_t n_points, int32_t F, int32_t H1, int32_t H2, int32_t H3) {
  (void)H3;
  if (n_points <= 0 || F <= 0 || H1 <= 0 || H2 <= 0) return 0;
  const int Kp = (F + kKTile - 1) / kKTile * kKTile;
  const int64_t rows = n_points >= kMaxChunkRows ? kMaxChunkRows : (n_points + kRowTile - 1) / kRowTile * kRowTile;
  return workspace_layout(rows, Kp, H1, H2).total;
}

int64_t list_query_chunk_rows(size_t workspace_bytes, int64_t n_points, int32_t F, int32_t H1, int32_t H2,
                              int32_t H3) {
  (void)H3;
  if (n_points <= 0 || F <= 0 || H1 <= 0 || H2 <= 0) return 0;
  return chunk_rows_for(workspace_bytes, n_points, (F + kKTile - 1) / kKTile * kKTile, H1, H2);
}

// ------------------------------------------------------------------------------------------ projected perceptual map
static int64_t proj_rows(int32_t B, int32_t map_size) {
  const int64_t px = (int64_t)B * map_size * map_size;
  return (px + kRowTile - 1) / kRowTile * kRowTile;
}

size_t list_percep_proj_bytes(int32_t B, int32_t map_size, int32_t H1, int32_t precision) {
  if (B <= 0 || map_size < 2 || H1 <= 0) return 0;
  return (size_t)proj_rows(B, map_size) * H1 * (precision == LIST_PREC_FP16 ? 2 : 4);
}

size_t list_percep_proj_scratch_bytes(int32_t B, int32_t map_size, int32_t img_C, int32_t precision) {
  if (B <= 0 || map_size < 2 || img_C <= 0) return 0;
  if (precision == LIST_PREC_FP16) return 0;
  return (size_t)B * map_size * map_size * img_C * 4;          // bf16 hi + lo halfs of the fp32 map, interleaved
}

int list_prep_percep_proj(const void* img_map, int32_t img_dtype, int32_t B, int32_t map_size,
                          const int32_t vox_C[LIST_N_VOX_LEVELS], int32_t img_C, const void* packed_mlp,
                          int32_t H1, int32_t H2, int32_t H3, int32_t precision, void* proj,
                          size_t proj_bytes, void* scratch, size_t scratch_bytes, void* stream) {
  if (!img_map || !vox_C || !packed_mlp || !proj) return fail(LIST_ERR_ARG, "NULL pointer");
  if (precision < LIST_PREC_BF16X3 || precision > LIST_PREC_FP16) return fail(LIST_ERR_ARG, "precision=%d", precision);
  if (B <= 0 || map_size < 2) return fail(LIST_ERR_SHAPE, "B=%d map_size=%d", B, map_size);
  FeatLayout L;
  if (!make_layout(vox_C, img_C, &L)) return fail(LIST_ERR_UNSUPPORTED, "unsupported channel counts");
  const bool fp16 = precision == LIST_PREC_FP16;
  if (img_C % 64 || H1 % 256 || H1 <= 0)
    return fail(LIST_ERR_UNSUPPORTED, "percep_proj needs img_C %% 64 == 0 and H1 %% 256 == 0 (img_C=%d, H1=%d)", img_C, H1);
  if (fp16 != (img_dtype == LIST_MAP_F16))
    return fail(LIST_ERR_UNSUPPORTED, "percep_proj: fp16 operands pair with fp16 maps, the bf16 formats with fp32 maps");
  if (!aligned16(img_map) || !aligned16(packed_mlp) || !aligned16(proj) || (scratch && !aligned16(scratch)))
    return fail(LIST_ERR_SHAPE, "buffers must be 16-byte aligned");
  if (proj_bytes < list_percep_proj_bytes(B, map_size, H1, precision))
    return fail(LIST_ERR_WORKSPACE, "proj buffer too small: %zu < %zu", proj_bytes, list_percep_proj_bytes(B, map_size, H1, precision));
  const size_t need = list_percep_proj_scratch_bytes(B, map_size, img_C, precision);
  if (need && (!scratch || scratch_bytes < need))
    return fail(LIST_ERR_WORKSPACE, "scratch too small: %zu < %zu", scratch_bytes, need);
  const PackedMlp pk = packed_mlp_layout(L.Kp, H1, H2, H3);
  if (!fp16 && pk.w0_lo != pk.w0_hi + (size_t)H1 * L.Kp * 2)
    return fail(LIST_ERR_ARG, "internal: hi / lo planes of the packed fc_0 weight are not contiguous");
  hipStream_t s = (hipStream_t)stream;
  const int64_t px = (int64_t)B * map_size * map_size;
  const char* a_ptr = (const char*)img_map;
  if (!fp16) {
    hipError_t e = launch_split_xi((const float*)img_map, (unsigned short*)scratch, px * img_C, s);
    if (e != hipSuccess) return hip_fail(e, "map split launch");
    a_ptr = (const char*)scratch;
  }
  // proj[pixel][n] = sum_c map[pixel][c] * W0[n][c]: the perceptual block is the FIRST img_C columns of the packed weight
  GemmParams gp;
  memset(&gp, 0, sizeof(gp));
  gp.fmt = fp16 ? FMT_FP16 : FMT_BF16_SPLIT;
  gp.x3i = fp16 ? 0 : 1;
  gp.a_hi = a_ptr; gp.a_lo = a_ptr;
  gp.w_hi = (const char*)packed_mlp + pk.w0_hi; gp.w_lo = gp.w_hi;
  gp.M = (int)proj_rows(B, map_size); gp.N = H1; gp.K = img_C;
  gp.lda = img_C; gp.ldw = L.Kp; gp.a_rows = (int)px;
  gp.dx = proj; gp.dx_f16 = fp16 ? 1 : 0; gp.n_store = H1; gp.ldo = H1;
  hipError_t e = launch_gemm(gp, precision == LIST_PREC_BF16X3 ? 3 : 1, EPI_DX, s);
  if (e != hipSuccess) return hip_fail(e, "projection launch");
  return LIST_OK;
}

// ------------------------------------------------------------------------------------------ projected encoder levels
namespace {
struct ImgProjPlan {
  int kept_C, img_C, n_proj;
  int64_t rows[LIST_N_IMG_LEVELS], rows_pad[LIST_N_IMG_LEVELS];        // B * H * W of a projected level (padded to 256)
  size_t a_f32[LIST_N_IMG_LEVELS], a_op[LIST_N_IMG_LEVELS], p[LIST_N_IMG_LEVELS];   // scratch offsets: rows (fp32 staging, split
  size_t scratch;                                                      //   formats only), operand rows, projected level
};
int img_proj_plan(const ListMap2D* maps, int32_t B, int32_t n_kept, int32_t H1, int32_t precision, ImgProjPlan* pl) {
  if (!maps) return fail(LIST_ERR_ARG, "maps is NULL");
  if (precision < LIST_PREC_BF16X3 || precision > LIST_PREC_FP16) return fail(LIST_ERR_ARG, "precision=%d", precision);
  if (B <= 0 || B > 65535) return fail(LIST_ERR_SHAPE, "B=%d", B);
  if (n_kept < 0 || n_kept >= LIST_N_IMG_LEVELS)
    return fail(LIST_ERR_ARG, "n_kept_levels=%d (need 0 .. %d: at least one level is projected)", n_kept, LIST_N_IMG_LEVELS - 1);
  if (H1 <= 0 || H1 % 256) return fail(LIST_ERR_UNSUPPORTED, "H1=%d (need a multiple of 256)", H1);
  const bool fp16 = precision == LIST_PREC_FP16;
  pl->kept_C = 0; pl->img_C = 0; pl->n_proj = LIST_N_IMG_LEVELS - n_kept;
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t r = o; o = align_up(o + bytes, 256); return r; };
  for (int i = 0; i < LIST_N_IMG_LEVELS; ++i) {
    const ListMap2D& m = maps[i];
    if (!m.data || m.C < 1 || m.H < 1 || m.W < 1) return fail(LIST_ERR_SHAPE, "image level %d: bad descriptor", i);
    if (m.C % 64) return fail(LIST_ERR_UNSUPPORTED, "image level %d: C=%d (img_proj needs multiples of 64)", i, m.C);
    pl->img_C += m.C;
    if (i < n_kept) { pl->kept_C += m.C; continue; }
    const int64_t rows = (int64_t)B * m.H * m.W;
    if (rows * (m.C > H1 ? m.C : H1) >= (int64_t)1 << 31) return fail(LIST_ERR_SHAPE, "image level %d: too large to project", i);
    pl->rows[i] = rows; pl->rows_pad[i] = (rows + kRowTile - 1) / kRowTile * kRowTile;
    pl->a_f32[i] = fp16 ? 0 : take((size_t)rows * m.C * 4);
    pl->a_op[i] = take((size_t)rows * m.C * (fp16 ? 2 : 4));
    pl->p[i] = take((size_t)pl->rows_pad[i] * H1 * (fp16 ? 2 : 4));       // projected level: halfs for fp16 operands
  }
  pl->scratch = o;
  return LIST_OK;
}
}  // namespace

size_t list_img_proj_map_bytes(const ListMap2D maps[LIST_N_IMG_LEVELS], int32_t B, int32_t map_size,
                               int32_t n_kept_levels, int32_t H1, int32_t precision) {
  ImgProjPlan pl;
  if (map_size < 2 || img_proj_plan(maps, B, n_kept_levels, H1, precision, &pl) != LIST_OK) return 0;
  return (size_t)B * map_size * map_size * (pl.kept_C + H1) * (precision == LIST_PREC_FP16 ? 2 : 4);
}

size_t list_img_proj_scratch_bytes(const ListMap2D maps[LIST_N_IMG_LEVELS], int32_t B, int32_t n_kept_levels,
                                   int32_t H1, int32_t precision) {
  ImgProjPlan pl;
  if (img_proj_plan(maps, B, n_kept_levels, H1, precision, &pl) != LIST_OK) return 0;
  return pl.scratch;
}

int list_prep_img_proj(const ListMap2D maps[LIST_N_IMG_LEVELS], int32_t B, int32_t map_size,
                       int32_t n_kept_levels, const int32_t vox_C[LIST_N_VOX_LEVELS], const void* packed_mlp,
                       int32_t H1, int32_t H2, int32_t H3, int32_t precision, void* out, size_t out_bytes,
                       void* scratch, size_t scratch_bytes, void* stream) {
  if (!vox_C || !packed_mlp || !out || !scratch) return fail(LIST_ERR_ARG, "NULL pointer");
  if (map_size < 2 || map_size > 320) return fail(LIST_ERR_SHAPE, "map_size=%d (need 2..320)", map_size);
  ImgProjPlan pl;
  int rc = img_proj_plan(maps, B, n_kept_levels, H1, precision, &pl);
  if (rc != LIST_OK) return rc;
  FeatLayout L;
  if (!make_layout(vox_C, pl.img_C, &L)) return fail(LIST_ERR_UNSUPPORTED, "unsupported channel counts");
  if (L.img_off != 0) return fail(LIST_ERR_ARG, "internal: the perceptual block does not lead the feature layout");
  const bool fp16 = precision == LIST_PREC_FP16;
  if (!aligned16(packed_mlp) || !aligned16(out) || !aligned16(scratch)) return fail(LIST_ERR_SHAPE, "buffers must be 16-byte aligned");
  const size_t need_out = list_img_proj_map_bytes(maps, B, map_size, n_kept_levels, H1, precision);
  if (out_bytes < need_out) return fail(LIST_ERR_WORKSPACE, "out buffer too small: %zu < %zu", out_bytes, need_out);
  if (scratch_bytes < pl.scratch) return fail(LIST_ERR_WORKSPACE, "scratch too small: %zu < %zu", scratch_bytes, pl.scratch);
  if ((int64_t)map_size * map_size * (pl.kept_C + H1) >= (int64_t)1 << 31) return fail(LIST_ERR_SHAPE, "projected map larger than 2^31 elements");
  const PackedMlp pk = packed_mlp_layout(L.Kp, H1, H2, H3);
  if (!fp16 && pk.w0_lo != pk.w0_hi + (size_t)H1 * L.Kp * 2)
    return fail(LIST_ERR_ARG, "internal: hi / lo planes of the packed fc_0 weight are not contiguous");
  hipStream_t s = (hipStream_t)stream;
  const int Ct = pl.kept_C + H1;
  hipError_t e = launch_prep_img(maps, B, map_size, Ct, fp16 ? 1 : 0, out, s, n_kept_levels);
  if (e != hipSuccess) return hip_fail(e, "prep_img launch");
  ListMap2D proj[LIST_N_IMG_LEVELS];
  int coff = pl.kept_C, n_proj = 0;
  char* sc = (char*)scratch;
  // rows [B*H*W][C] of every projected level in the operand format (fp16, or bf16 hi / lo interleaved through an fp32
  // staging copy): one launch
  {
    void* outs[LIST_N_IMG_LEVELS];
    for (int i = n_kept_levels; i < LIST_N_IMG_LEVELS; ++i) outs[i - n_kept_levels] = fp16 ? sc + pl.a_op[i] : sc + pl.a_f32[i];
    e = launch_img_level_rows(maps + n_kept_levels, outs, pl.n_proj, B, fp16 ? 1 : 0, s);
    if (e != hipSuccess) return hip_fail(e, "level rows launch");
  }
  // P_l[pixel][n] = sum_c rows[pixel][c] * W0[n][coff_l + c]: every level in ONE grouped launch of the ping-pong kernel
  // (fp16 operands: P_l in halfs, like the map it is resized into; bf16 formats: fp32)
  GemmParams gp;
  memset(&gp, 0, sizeof(gp));
  gp.fmt = fp16 ? FMT_FP16 : FMT_BF16_SPLIT;
  gp.x3i = fp16 ? 0 : 1;
  // fp16 operands keep the projected levels in halfs, like the map they are resized into (measured against fp32 levels,
  // round 4b: the SDF error of config 2 and of the golden cases moves by its own noise, 5.2e-5 / 5.3e-5 against the
  // standard path; the resize reads half the bytes, prep 0.176 -> 0.171 ms)
  const int p16 = fp16 ? 1 : 0;
  gp.N = H1; gp.ldw = L.Kp; gp.dx_f16 = p16; gp.n_store = H1; gp.ldo = H1;
  if (pl.n_proj > kGemmMaxGroups) return fail(LIST_ERR_UNSUPPORTED, "at most %d levels can be projected (n_kept_levels >= %d)", kGemmMaxGroups, LIST_N_IMG_LEVELS - kGemmMaxGroups);
  int64_t m_total = 0;
  for (int i = n_kept_levels; i < LIST_N_IMG_LEVELS; ++i) {
    const ListMap2D& m = maps[i];
    if (!fp16) {
      e = launch_split_xi((const float*)(sc + pl.a_f32[i]), (unsigned short*)(sc + pl.a_op[i]), pl.rows[i] * m.C, s);
      if (e != hipSuccess) return hip_fail(e, "level split launch");
    }
    GemmGroup& gr = gp.grp[gp.n_groups++];
    gr.a = sc + pl.a_op[i];
    gr.w = (const char*)packed_mlp + pk.w0_hi + (size_t)coff * (fp16 ? 2 : 4);
    gr.out = sc + pl.p[i]; gr.K = m.C; gr.a_rows = (int)pl.rows[i]; gr.m_tiles = (int)(pl.rows_pad[i] / kRowTile); gr.lda = m.C;
    m_total += pl.rows_pad[i];
    ListMap2D& pm = proj[n_proj++];
    pm.data = (const float*)(sc + pl.p[i]); pm.C = H1; pm.H = m.H; pm.W = m.W;
    pm.sc = 1; pm.sw = H1; pm.sh = (int64_t)m.W * H1; pm.sb = (int64_t)m.H * m.W * H1;
    coff += m.C;
  }
  if (m_total >= (int64_t)1 << 31) return fail(LIST_ERR_SHAPE, "projected levels too large");
  gp.M = (int)m_total;
  gp.a_hi = gp.grp[0].a; gp.a_lo = gp.a_hi; gp.w_hi = gp.grp[0].w; gp.w_lo = gp.w_hi; gp.dx = gp.grp[0].out;
  gp.K = gp.grp[0].K; gp.lda = gp.grp[0].lda; gp.a_rows = gp.grp[0].a_rows;
  e = launch_gemm(gp, precision == LIST_PREC_BF16X3 ? 3 : 1, EPI_DX, s);
  if (e != hipSuccess) return hip_fail(e, "level projection launch");
  e = launch_proj_resize_sum(proj, n_proj, B, map_size, Ct, pl.kept_C, fp16 ? 1 : 0, out, s, p16);
  if (e != hipSuccess) return hip_fail(e, "projected resize launch");
  return LIST_OK;
}

int list_sdf_query_fwd(const ListQueryArgs* a, void* stream) {
  FeatLayout L;
  if (a && a->B >= 0 && a->N >= 0 && (int64_t)a->B * a->N == 0) return LIST_OK;   // empty query: nothing to do
  int rc = check_query_common(a, &L);
  if (rc != LIST_OK) return rc;
  if (!a->sdf || !a->packed_mlp) return fail(LIST_ERR_ARG, "sdf/packed_mlp is NULL");
  if (a->F != L.F) return fail(LIST_ERR_SHAPE, "F=%d but channels give %d", a->F, L.F);
  if (a->H1 % 256 || a->H2 % 256 || a->H3 != 256 || a->H1 <= 0 || a->H2 <= 0)
    return fail(LIST_ERR_UNSUPPORTED, "hidden sizes %d/%d/%d", a->H1, a->H2, a->H3);
  if (a->precision != LIST_PREC_BF16X3 && a->precision != LIST_PREC_BF16 &&
      a->precision != LIST_PREC_FP16)
    return fail(LIST_ERR_ARG, "precision=%d", a->precision);
  // (a flag, not a count: anything else is a struct of another ABI version or an uninitialised one)
  if (a->no_fused_fc0 != 0 && a->no_fused_fc0 != 1)
    return fail(LIST_ERR_ARG, "no_fused_fc0=%d (0 or 1; zero-initialise ListQueryArgs, list_abi_version() = %d)",
                a->no_fused_fc0, LIST_ABI_VERSION);
  if (a->no_activations != 0 && a->no_activations != 1)
    return fail(LIST_ERR_ARG, "no_activations=%d (0 or 1; zero-initialise ListQueryArgs, list_abi_version() = %d)",
                a->no_activations, LIST_ABI_VERSION);
  if (!aligned16(a->workspace)) return fail(LIST_ERR_SHAPE, "workspace must be 16-byte aligned");
  const int64_t P = (int64_t)a->B * a->N;
  const int64_t rows = chunk_rows_for(a->workspace_bytes, P, L.Kp, a->H1, a->H2);
  if (rows < kRowTile)
    return fail(LIST_ERR_WORKSPACE, "workspace of %zu bytes cannot hold one %d-row tile",
                a->workspace_bytes, kRowTile);
  const Workspace ws = workspace_layout(rows, L.Kp, a->H1, a->H2);
  const PackedMlp pk = packed_mlp_layout(L.Kp, a->H1, a->H2, a->H3);
  const char* wp = (const char*)a->packed_mlp;
  char* wsb = (char*)a->workspace;
  const int terms = a->precision == LIST_PREC_BF16X3 ? 3 : 1;
  hipStream_t s = (hipStream_t)stream;

  // stage events: one set of LIST_N_STAGES per row chunk (a chunk beyond the caller's sets records nothing);
  // the gather launcher marks through the same (per-chunk) array
  ListQueryArgs chunk_args = *a;
  const int sets = a->stage_events ? (a->stage_event_sets > 0 ? a->stage_event_sets : 1) : 0;
  int chunk = 0;
  for (int64_t p0 = 0; p0 < P; p0 += rows, ++chunk) {
    const int n_valid = (int)((P - p0 < rows) ? (P - p0) : rows);
    const int crow = (n_valid + kRowTile - 1) / kRowTile * kRowTile;
    GatherParams g = make_gather(a, L, ws, p0, n_valid, crow);
    void* const* events = chunk < sets ? a->stage_events + (size_t)chunk * LIST_N_STAGES : nullptr;
    chunk_args.stage_events = events;
    auto mark = [&](int stage) {
      if (events && events[stage]) (void)hipEventRecord((hipEvent_t)events[stage], s);
    };
    mark(LIST_STAGE_BEGIN);
    hipError_t e = hipSuccess;
    if (!a->no_sort) {
      SortBuffers sb;
      // the pixel order serves the 2-D gather kernel (and, behind a training forward, the backward's map-side gather):
      // a forward whose fc_0 samples the map itself (k_fc0_fused) sorts by Morton cell only -- no projection per point,
      // half the counters, one scan and two index arrays less
      const bool want_pix = !takes_fused_fc0(a, L);
      sb.order = (int*)(wsb + ws.order); sb.order_img = want_pix ? (int*)(wsb + ws.order_img) : nullptr;
      sb.row_of = (int*)(wsb + ws.row_of); sb.keys = (int*)(wsb + ws.keys);
      sb.keys2 = (int*)(wsb + ws.keys2); sb.bins = (int*)(wsb + ws.bins);
      e = launch_sort_points(g, *a, sb, s);
      if (e != hipSuccess) return hip_fail(e, "sort launch");
      g.order = sb.order;
      if (want_pix && a->percep_feat == nullptr && a->map_size * ((a->map_size + 3) / 4) <= kSortPixCells) {
        g.order_img = sb.order_img; g.row_of = sb.row_of;
      }
    }
    mark(LIST_STAGE_SORT);
    int* nan_tiles = (int*)(wsb + ws.nan_tiles);
    const bool fused0 = takes_fused_fc0_any(a, L);            // the 128 x 512 kernel runs
    e = launch_gather(g, L, chunk_args, nan_tiles, s, /*skip_img=*/takes_fused_fc0(a, L));
    if (e != hipSuccess) return hip_fail(e, "gather launch");
    mark(LIST_STAGE_TAIL);

    GemmParams gp;
    memset(&gp, 0, sizeof(gp));
    gp.fmt = g.fmt;
    // fc_0 + ReLU
    gp.a_hi = wsb + ws.x_hi; gp.a_lo = wsb + ws.x_lo;
    gp.w_hi = wp + pk.w0_hi; gp.w_lo = wp + pk.w0_lo;
    gp.bias = (const float*)(wp + pk.b0);
    gp.M = crow; gp.N = a->H1; gp.K = L.Kp;
    // bf16 split formats: X and the packed fc_0 weight hold their hi / lo halfs interleaved (list_common.h xi_off)
    gp.x3i = g.fmt == FMT_FP16 ? 0 : 1;
    if (gp.x3i && (ws.x_lo != ws.x_hi + (size_t)rows * L.Kp * 2 || pk.w0_lo != pk.w0_hi + (size_t)a->H1 * L.Kp * 2))
      return fail(LIST_ERR_ARG, "internal: hi / lo planes of X or of the packed fc_0 weight are not contiguous");
    if (a->percep_proj) {
      // the perceptual block (the first img_C columns) is left out of the K loop; its contribution was sampled from
      // the projected map into the head of every X row and is added in the epilogue
      const int64_t skip = (int64_t)a->img_C * (gp.x3i ? 4 : 2);
      gp.a_hi += skip; gp.w_hi += skip;
      gp.K = L.Kp - a->img_C; gp.lda = L.Kp; gp.ldw = L.Kp;
      gp.rowvec = wsb + ws.x_hi; gp.rowvec_stride = (int64_t)L.Kp * (gp.x3i ? 4 : 2);
    } else if (a->img_proj) {
      // the projected levels' columns (img_kept_C .. img_C of the perceptual block, which leads the rows) are left out
      // of the K loop: their contribution is the row vector the 2-D gather sampled from the projected channels
      const int ktile = gp.x3i ? 32 : 64;                   // columns per 128-byte K-tile
      gp.k_gap_at = a->img_kept_C / ktile; gp.k_gap = (a->img_C - a->img_kept_C) / ktile;
      gp.K = L.Kp - (a->img_C - a->img_kept_C); gp.lda = L.Kp; gp.ldw = L.Kp;
      gp.rowvec = (const char*)g.rowvec; gp.rowvec_stride = g.rv_stride * 4;
      if (a->H1 * 4 > L.Kp * 2) return fail(LIST_ERR_UNSUPPORTED, "img_proj: H1 * 4 bytes exceed a row of the feature matrix's lo plane");
    }
    const char* rowvec_of_call = gp.rowvec;
    gp.out_hi = (unsigned short*)(wsb + ws.h1_hi);
    gp.out_lo = terms == 3 ? (unsigned short*)(wsb + ws.h1_lo) : nullptr;
    gp.ldo = a->H1;
    // exact border semantics (gather_kernels.hip): fc_0 flags the row tiles whose output holds a NaN, the gathers
    // of those tiles are redone with the reference's skip semantics and fc_0 runs again for them.  On finite
    // inputs both gated launches exit at their first instruction.
    gp.nan_tiles = nan_tiles;
    if (fused0) {
      // the perceptual block of X is produced inside fc_0 (fused_fc0_kernels.hip): no 2-D gather launch above, those
      // columns of X stay unwritten (the exact redo below rewrites them for the tiles it flags)
      FusedFc0Params fp;
      fp.gp = gp; fp.g = g; fp.img_map = a->img_map; fp.trans_mat = a->trans_mat;
      fp.ms = a->map_size; fp.Ct = a->img_C; fp.clamp_hi = a->clamp_hi; fp.n_produced = fused_produced_tiles(a);
      fp.proj = 0; fp.kept = 0;
      if (a->img_proj) {
        // the kernel samples the projected channels itself (no row-vector buffer); the exact redo below takes the row
        // vectors the fix-up kernel writes for the tiles it flags
        fp.proj = 1; fp.kept = a->img_kept_C; fp.Ct = a->img_kept_C + a->H1; fp.gp.rowvec = nullptr;
      }
      if (!fused_fc0_eligible(fp.gp, a->img_dtype == LIST_MAP_F16, a->img_proj ? a->img_kept_C : a->img_C))
        return fail(LIST_ERR_ARG, "internal: fused fc_0 taken for arguments it does not support");
      e = launch_fc0_fused(fp, terms, s);
    } else {
      e = launch_gemm(gp, terms, EPI_RELU_SPLIT, s);
    }
    if (e != hipSuccess) return hip_fail(e, "fc_0 launch");
    mark(LIST_STAGE_FC0);
    e = launch_gather_fixup(g, L, chunk_args, nan_tiles, s);
    if (e != hipSuccess) return hip_fail(e, "gather fix-up launch");
    gp.nan_tiles = nullptr; gp.tile_gate = nan_tiles; gp.rowvec = rowvec_of_call;
    e = launch_gemm(gp, terms, EPI_RELU_SPLIT, s);
    if (e != hipSuccess) return hip_fail(e, "gated fc_0 launch");
    gp.tile_gate = nullptr; gp.x3i = 0;
    gp.lda = gp.ldw = 0; gp.rowvec = nullptr; gp.k_gap_at = gp.k_gap = 0;
    mark(LIST_STAGE_EXACT);
    // fc_1 + ReLU
    gp.a_hi = wsb + ws.h1_hi; gp.a_lo = wsb + ws.h1_lo;
    gp.w_hi = wp + pk.w1_hi; gp.w_lo = wp + pk.w1_lo;
    gp.bias = (const float*)(wp + pk.b1);
    gp.N = a->H2; gp.K = a->H1;
    gp.out_hi = (unsigned short*)(wsb + ws.h2_hi);
    gp.out_lo = terms == 3 ? (unsigned short*)(wsb + ws.h2_lo) : nullptr;
    gp.ldo = a->H2;
    if (takes_fused_tail(a)) {
      // inference: fc_1, fc_2 and fc_out in one kernel, H2 stays in registers (gemm_kernels.hip, k_mlp_tail_f16)
      mark(LIST_STAGE_FC1);
      e = launch_mlp_tail(gp, wp + pk.w2_hi, (const float*)(wp + pk.b2), (const float*)(wp + pk.w3),
                          (const float*)(wp + pk.b3), a->sdf + p0, g.order, n_valid, s);
      if (e != hipSuccess) return hip_fail(e, "fc_1/fc_2/fc_out launch");
      mark(LIST_STAGE_FC2);
      continue;
    }
    e = launch_gemm(gp, terms, EPI_RELU_SPLIT, s);
    if (e != hipSuccess) return hip_fail(e, "fc_1 launch");
    mark(LIST_STAGE_FC1);
    // fc_2 + ReLU + fc_out
    gp.a_hi = wsb + ws.h2_hi; gp.a_lo = wsb + ws.h2_lo;
    gp.w_hi = wp + pk.w2_hi; gp.w_lo = wp + pk.w2_lo;
    gp.bias = (const float*)(wp + pk.b2);
    gp.N = a->H3; gp.K = a->H2;
    // H3 is kept with H1 / H2 for list_sdf_query_bwd (its head needs relu(fc_2): mask, dZ3, d fc_out.weight) -- by
    // forwards a backward can follow only: an inference forward that misses the fused tail above (bf16 formats, the
    // 256^3 grid of config 4 among them) would write 1 KB per point that nobody reads
    gp.out_hi = a->no_activations ? nullptr : (unsigned short*)(wsb + ws.h3_hi);
    gp.out_lo = (terms == 3 && !a->no_activations) ? (unsigned short*)(wsb + ws.h3_lo) : nullptr;
    gp.ldo = a->H3;
    gp.w3 = (const float*)(wp + pk.w3); gp.b3 = (const float*)(wp + pk.b3);
    gp.sdf = a->sdf + p0; gp.n_valid = n_valid; gp.order = g.order;
    e = launch_gemm(gp, terms, EPI_RELU_DOT, s);
    if (e != hipSuccess) return hip_fail(e, "fc_2/fc_out launch");
    mark(LIST_STAGE_FC2);
  }
  return LIST_OK;
}

int list_query_plan(const ListQueryArgs* a, ListQueryPlan* plan) {
  if (!plan) return fail(LIST_ERR_ARG, "plan is NULL");
  memset(plan, 0, sizeof(*plan));
  FeatLayout L;
  if (a && a->B >= 0 && a->N >= 0 && (int64_t)a->B * a->N == 0) return LIST_OK;   // empty query: no launch at all
  int rc = check_query_common(a, &L);
  if (rc != LIST_OK) return rc;
  if (a->H1 <= 0 || a->H2 <= 0) return fail(LIST_ERR_UNSUPPORTED, "hidden sizes %d/%d/%d", a->H1, a->H2, a->H3);
  const int64_t P = (int64_t)a->B * a->N;
  const int64_t rows = chunk_rows_for(a->workspace_bytes, P, L.Kp, a->H1, a->H2);
  if (rows < kRowTile) return fail(LIST_ERR_WORKSPACE, "workspace of %zu bytes cannot hold one %d-row tile",
                                   a->workspace_bytes, kRowTile);
  plan->rows_per_chunk = rows;
  plan->chunks = (int32_t)((P + rows - 1) / rows);
  plan->fused_tail = takes_fused_tail(a) ? 1 : 0;
  plan->fc0_k = a->percep_proj ? L.Kp - a->img_C : a->img_proj ? L.Kp - (a->img_C - a->img_kept_C) : L.Kp;
  plan->img_proj = a->img_proj ? 1 : 0;
  {
    const Workspace ws = workspace_layout(rows, L.Kp, a->H1, a->H2);
    const int n_valid = (int)(P < rows ? P : rows);
    const GatherParams g = make_gather(a, L, ws, 0, n_valid, (n_valid + kRowTile - 1) / kRowTile * kRowTile);
    plan->box_levels = gather_box_levels(g, L, *a);
    plan->fused_fc0 = takes_fused_fc0(a, L) ? 1 : 0;
  }
  return LIST_OK;
}

int list_gather_features_fwd(const ListQueryArgs* a, float* out, void* stream) {
  FeatLayout L;
  int rc = check_query_common(a, &L);
  if (rc != LIST_OK) return rc;
  if (!out) return fail(LIST_ERR_ARG, "out is NULL");
  if (a->percep_proj || a->img_proj) return fail(LIST_ERR_UNSUPPORTED, "percep_proj / img_proj leave perceptual features out of X");
  const int64_t P = (int64_t)a->B * a->N;
  const int H1 = a->H1 > 0 ? a->H1 : 512, H2 = a->H2 > 0 ? a->H2 : 256;
  const int64_t rows = chunk_rows_for(a->workspace_bytes, P, L.Kp, H1, H2);
  if (rows < kRowTile) return fail(LIST_ERR_WORKSPACE, "workspace too small");
  const Workspace ws = workspace_layout(rows, L.Kp, H1, H2);
  hipStream_t s = (hipStream_t)stream;
  for (int64_t p0 = 0; p0 < P; p0 += rows) {
    const int n_valid = (int)((P - p0 < rows) ? (P - p0) : rows);
    const int crow = (n_valid + kRowTile - 1) / kRowTile * kRowTile;
    GatherParams g = make_gather(a, L, ws, p0, n_valid, crow);
    int* nan_tiles = (int*)((char*)a->workspace + ws.nan_tiles);
    hipError_t e = launch_gather(g, L, *a, nan_tiles, s);
    if (e != hipSuccess) return hip_fail(e, "gather launch");
    e = launch_features_out(g, L, out, nan_tiles, s);
    if (e != hipSuccess) return hip_fail(e, "features_out launch");
    e = launch_gather_fixup(g, L, *a, nan_tiles, s);          // exact redo of the tiles that showed a NaN
    if (e != hipSuccess) return hip_fail(e, "gather fix-up launch");
    e = launch_features_out(g, L, out, nullptr, s);
    if (e != hipSuccess) return hip_fail(e, "features_out launch");
  }
  return LIST_OK;
}

int list_percep_pool_fwd(const ListPoolArgs* a, void* stream) {
  if (!a) return fail(LIST_ERR_ARG, "args is NULL");
  if (!a->pc || !a->trans_mat || !a->img_map || !a->out) return fail(LIST_ERR_ARG, "NULL pointer");
  if (a->B <= 0 || a->N <= 0 || a->map_size < 2 || a->img_C <= 0 || a->img_C % 4)
    return fail(LIST_ERR_SHAPE, "B=%d N=%d map_size=%d img_C=%d", a->B, a->N, a->map_size, a->img_C);
  if (!aligned16(a->img_map)) return fail(LIST_ERR_SHAPE, "img_map must be 16-byte aligned");
  if (!dtype_ok(a->img_dtype) || (a->img_dtype == LIST_MAP_F16 && a->img_C % 8))
    return fail(LIST_ERR_ARG, "img_dtype=%d img_C=%d", a->img_dtype, a->img_C);
  if ((int64_t)a->map_size * a->map_size * a->img_C >= (int64_t)1 << 31)
    return fail(LIST_ERR_SHAPE, "image map larger than 2^31 elements");
  if (!(a->clamp_hi >= 0.f)) return fail(LIST_ERR_ARG, "clamp_hi=%g must be >= 0", (double)a->clamp_hi);
  hipError_t e = launch_percep_pool(*a, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "percep_pool launch");
  return LIST_OK;
}

// ------------------------------------------------------------------------------------------ backward
size_t list_packed_mlp_bwd_bytes(const ListMlpWeights* w) {
  FeatLayout L;
  if (weights_layout(w, &L) != LIST_OK) return 0;
  return packed_mlp_bwd_layout(L.Kp, w->H1, w->H2, w->H3).total;
}

int list_prep_mlp_weights_bwd(const ListMlpWeights* w, void* packed, size_t packed_bytes, void* stream) {
  FeatLayout L;
  int rc = weights_layout(w, &L);
  if (rc != LIST_OK) return rc;
  const PackedMlpBwd P = packed_mlp_bwd_layout(L.Kp, w->H1, w->H2, w->H3);
  if (!packed || packed_bytes < P.total)
    return fail(LIST_ERR_WORKSPACE, "packed_bwd buffer too small: %zu < %zu", packed_bytes, P.total);
  if (!aligned16(packed)) return fail(LIST_ERR_SHAPE, "packed_bwd must be 16-byte aligned");
  hipError_t e = launch_prep_weights_bwd(*w, L, P, (char*)packed, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "prep_weights_bwd launch");
  return LIST_OK;
}

size_t list_query_bwd_workspace_bytes(int64_t n_points, int32_t F, int32_t H1, int32_t H2, int32_t H3,
                                      int32_t precision) {
  if (n_points <= 0 || n_points > kMaxChunkRows || F <= 0 || H1 <= 0 || H2 <= 0 || H3 <= 0) return 0;
  const int Kp = (F + kKTile - 1) / kKTile * kKTile;
  const int64_t rows = (n_points + kRowTile - 1) / kRowTile * kRowTile;
  return bwd_workspace_layout(rows, Kp, H1, H2, H3, precision == LIST_PREC_FP16).total;
}

int list_sdf_query_bwd(const ListQueryGradArgs* ga, void* stream) {
  if (!ga || !ga->fwd) return fail(LIST_ERR_ARG, "args/fwd is NULL");
  const ListQueryArgs* a = ga->fwd;
  FeatLayout L;
  if (a->B >= 0 && a->N >= 0 && (int64_t)a->B * a->N == 0) return LIST_OK;
  int rc = check_query_common(a, &L);
  if (rc != LIST_OK) return rc;
  if (ga->grad_img_levels) {
    if (!ga->grad_img_map) return fail(LIST_ERR_ARG, "grad_img_levels needs grad_img_map as its intermediate");
    if (a->map_size > 320) return fail(LIST_ERR_SHAPE, "grad_img_levels: map_size=%d (need <= 320)", a->map_size);
    int ct = 0;
    for (int i = 0; i < LIST_N_IMG_LEVELS; ++i) {
      const ListMap2D& m = ga->grad_img_levels[i];
      if (m.C < 1 || m.H < 1 || m.W < 1) return fail(LIST_ERR_SHAPE, "grad_img_levels[%d]: bad descriptor", i);
      ct += m.C;
    }
    if (ct != a->img_C) return fail(LIST_ERR_SHAPE, "grad_img_levels: channels sum to %d, the map has %d", ct, a->img_C);
  }
  if (ga->grad_img_map_dtype != LIST_MAP_F32 && ga->grad_img_map_dtype != LIST_MAP_F16)
    return fail(LIST_ERR_ARG, "grad_img_map_dtype=%d", ga->grad_img_map_dtype);
  if (ga->grad_img_map_dtype == LIST_MAP_F16) {
    // halfs at the gradient scale between the map-side gather and the adjoint resize: an intermediate, not an output
    if (!ga->grad_img_levels || !ga->grad_img_map)
      return fail(LIST_ERR_ARG, "grad_img_map_dtype = F16 is the intermediate of grad_img_levels: pass both");
    if (a->precision != LIST_PREC_FP16)
      return fail(LIST_ERR_UNSUPPORTED, "grad_img_map_dtype = F16 needs fp16 operands (fwd->precision FP16)");
    if (a->no_sort || a->percep_feat || a->B > kSortImages || a->map_size * ((a->map_size + 3) / 4) > kSortPixCells)
      return fail(LIST_ERR_UNSUPPORTED, "grad_img_map_dtype = F16 needs the pixel-ordered gather form (point sort on, "
                                        "B <= %d, map_size * ceil(map_size / 4) <= %d)", kSortImages, kSortPixCells);
    if (a->img_C % 4) return fail(LIST_ERR_UNSUPPORTED, "grad_img_map_dtype = F16 needs img_C %% 4 == 0");
  }
  if (a->percep_feat && (ga->grad_img_map || ga->grad_trans_mat))
    return fail(LIST_ERR_ARG, "with percep_feat the perceptual gradient is grad_percep_feat, not grad_img_map/grad_trans_mat");
  if (!a->percep_feat && ga->grad_percep_feat)
    return fail(LIST_ERR_ARG, "grad_percep_feat needs the pre-pooled form (fwd->percep_feat)");
  if (!ga->grad_sdf || !ga->packed_mlp_bwd || !a->packed_mlp || !ga->workspace)
    return fail(LIST_ERR_ARG, "grad_sdf/packed_mlp_bwd/packed_mlp/workspace is NULL");
  if (a->F != L.F) return fail(LIST_ERR_SHAPE, "F=%d but channels give %d", a->F, L.F);
  if (a->H1 % 256 || a->H2 % 256 || a->H3 != 256 || a->H1 <= 0 || a->H2 <= 0)
    return fail(LIST_ERR_UNSUPPORTED, "hidden sizes %d/%d/%d", a->H1, a->H2, a->H3);
  if (a->H1 > 2048 || a->H2 > 2048)
    return fail(LIST_ERR_UNSUPPORTED, "backward supports hidden sizes up to 2048 (got %d/%d)", a->H1, a->H2);
  if (a->precision < LIST_PREC_BF16X3 || a->precision > LIST_PREC_FP16)
    return fail(LIST_ERR_ARG, "precision=%d", a->precision);
  if (ga->vox_adjoint < 0 || ga->vox_adjoint > 2) return fail(LIST_ERR_ARG, "vox_adjoint=%d", ga->vox_adjoint);
  if (a->percep_proj) return fail(LIST_ERR_UNSUPPORTED, "a forward with percep_proj (inference) keeps no perceptual features for the backward");
  if (a->img_proj) return fail(LIST_ERR_UNSUPPORTED, "a forward with img_proj (inference) keeps no perceptual features for the backward");
  if (a->no_activations) return fail(LIST_ERR_ARG, "a forward with no_activations = 1 (inference) keeps no H1 / H2 for the backward");
  const int64_t P = (int64_t)a->B * a->N;
  if (P > kMaxChunkRows)
    return fail(LIST_ERR_UNSUPPORTED, "backward handles up to %lld points per call (got %lld)",
                (long long)kMaxChunkRows, (long long)P);
  const int64_t rows = chunk_rows_for(a->workspace_bytes, P, L.Kp, a->H1, a->H2);
  if (rows < P)
    return fail(LIST_ERR_WORKSPACE, "forward workspace does not hold the whole query in one chunk");
  const bool fp16 = a->precision == LIST_PREC_FP16;
  const Workspace ws = workspace_layout(rows, L.Kp, a->H1, a->H2);
  const BwdWorkspace bw = bwd_workspace_layout(rows, L.Kp, a->H1, a->H2, a->H3, fp16);
  if (ga->workspace_bytes < bw.total)
    return fail(LIST_ERR_WORKSPACE, "backward workspace too small: %zu < %zu", ga->workspace_bytes, bw.total);
  if (!aligned16(ga->workspace)) return fail(LIST_ERR_SHAPE, "workspace must be 16-byte aligned");
  for (int l = 0; l < LIST_N_VOX_LEVELS; ++l) {
    const ListVoxLevel& gv = ga->grad_vox[l];
    if (!gv.data) continue;
    const ListVoxLevel& v = a->vox[l];
    if (gv.C != v.C || gv.D != v.D || gv.H != v.H || gv.W != v.W || gv.dtype != LIST_MAP_F32 ||
        gv.image_stride != (int64_t)v.C * v.D * v.H * v.W)
      return fail(LIST_ERR_SHAPE, "grad_vox[%d] must be fp32 [B][D][H][W][C] of the level's shape", l);
  }
  const PackedMlp pk = packed_mlp_layout(L.Kp, a->H1, a->H2, a->H3);
  const PackedMlpBwd pb = packed_mlp_bwd_layout(L.Kp, a->H1, a->H2, a->H3);
  const char* wp = (const char*)a->packed_mlp;
  const char* wt = (const char*)ga->packed_mlp_bwd;
  char* fw = (char*)a->workspace;
  char* bwp = (char*)ga->workspace;
  const int terms = a->precision == LIST_PREC_BF16X3 ? 3 : 1;
  const int fmt = fp16 ? FMT_FP16 : FMT_BF16_SPLIT;
  const bool lo = terms == 3;
  const int n_valid = (int)P;
  const int crow = (int)((P + kRowTile - 1) / kRowTile * kRowTile);
  hipStream_t s = (hipStream_t)stream;
  auto mark = [&](int stage) {
    if (ga->stage_events && ga->stage_events[stage])
      (void)hipEventRecord((hipEvent_t)ga->stage_events[stage], s);
  };
  hipError_t e = hipSuccess;
  // a failure after the fork must not leave the auxiliary streams running past the call: the caller frees the
  // workspace and the outputs as soon as it sees the error.  join_and_fail() orders `s` behind whatever has been
  // enqueued on them so far, then reports.
  hipStream_t j_direct = s, j_window = s, j_win2 = s;
  hipEvent_t ev_dw0 = nullptr;                    // dW0 done (forked calls that want d_trans_mat)
  auto join_and_fail = [&](hipError_t err, const char* what) -> int {
    if (ev_dw0) { (void)hipEventDestroy(ev_dw0); ev_dw0 = nullptr; }
    for (hipStream_t from : {j_direct, j_window, j_win2}) {
      if (from == s) continue;
      hipEvent_t ev;
      if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) continue;
      if (hipEventRecord(ev, from) == hipSuccess) (void)hipStreamWaitEvent(s, ev, 0);
      (void)hipEventDestroy(ev);
    }
    return hip_fail(err, what);
  };
#define LIST_TRY(call, what) do { e = (call); if (e != hipSuccess) return join_and_fail(e, what); } while (0)

  const int* order = a->no_sort ? nullptr : (const int*)(fw + ws.order);
  float* scale = (float*)(bwp + bw.scale);
  float* colsum = (float*)(bwp + bw.colsum);
  float* slab = (float*)(bwp + bw.slab);
  auto plane = [&](size_t off) { return (unsigned short*)(bwp + off); };

  // optional fork/join onto the caller's auxiliary streams (see ListQueryGradArgs.aux_streams)
  hipStream_t s_direct = s, s_window = s, s_win2 = s;
  const bool forked = ga->aux_streams[0] && ga->aux_streams[1];
  if (forked) { s_direct = (hipStream_t)ga->aux_streams[0]; s_window = (hipStream_t)ga->aux_streams[1]; }
  if (forked && ga->aux_streams[2]) s_win2 = (hipStream_t)ga->aux_streams[2];
  j_direct = s_direct; j_window = s_window; j_win2 = s_win2;
  auto hand_over = [&](hipStream_t from, hipStream_t to) -> hipError_t {      // `to` continues after `from`'s work so far
    if (from == to) return hipSuccess;
    hipEvent_t ev;
    hipError_t err = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    if (err != hipSuccess) return err;
    err = hipEventRecord(ev, from);
    if (err == hipSuccess) err = hipStreamWaitEvent(to, ev, 0);
    (void)hipEventDestroy(ev);                    // released once the wait has been satisfied
    return err;
  };
  // The bias / fc_out.weight column sums feed nothing downstream: forked, they leave the chain of dependent GEMMs
  // and run on the atomics' stream, which has nothing to do until dX exists (same kernels, same order among
  // themselves -- they share the `colsum` scratch).
  auto side_colsum = [&](const unsigned short* z_hi, const unsigned short* z_lo, int N, const float* gsdf,
                         const int* ord, int use_inv_scale, float* out) -> hipError_t {
    hipStream_t to = s;
    if (forked) {
      const hipError_t err = hand_over(s, s_direct);
      if (err != hipSuccess) return err;
      to = s_direct;
    }
    return launch_colsum(z_hi, z_lo, crow, n_valid, N, fmt, gsdf, ord, scale, use_inv_scale, colsum, out, to);
  };
  ScatterParams sp;
  sp.g = make_gather(a, L, ws, 0, n_valid, crow);
  sp.g.order = order;
  const bool pix = !a->no_sort && !a->percep_feat && a->map_size * ((a->map_size + 3) / 4) <= kSortPixCells;
  if (pix) { sp.g.order_img = (const int*)(fw + ws.order_img); sp.g.row_of = (const int*)(fw + ws.row_of); }
  sp.dx = bwp + bw.dx; sp.dx_f16 = fp16 ? 1 : 0; sp.scale = scale; sp.forked = forked ? 1 : 0;
  VoxGatherBuffers vb;
  vb.keys = (int*)(bwp + bw.vs_keys); vb.bins = (int*)(bwp + bw.vs_bins); vb.sums = (int*)(bwp + bw.vs_sums);
  vb.recs = bwp + bw.vs_recs; vb.mode = ga->vox_adjoint;
  // fp16 operands leave the lo planes of the backward workspace unused: the dZ1 lo plane (rows x H1 halfs) is the
  // scratch of the packed-half atomics (a level whose fp16 image does not fit it keeps the fp32 atomics)
  vb.h16 = fp16 ? (void*)(bwp + bw.dz1_lo) : nullptr;
  vb.h16_bytes = fp16 ? (size_t)rows * a->H1 * 2 : 0;
  vb.h16w = fp16 ? (void*)(bwp + bw.dz2_lo) : nullptr;               // window levels: the dZ2 lo plane
  vb.h16w_bytes = fp16 ? (size_t)rows * a->H2 * 2 : 0;
  const ScatterStreams sst = {s, s_direct, s_window, s_win2};

  mark(LIST_BWD_BEGIN);
  // --- head: scale, dZ3, d fc_out (H3 = relu(fc_2) as the forward left it in its workspace) -------------------
  LIST_TRY(launch_grad_scale(ga->grad_sdf, P, fp16 ? 1 : 0, scale, ga->mlp.b3, colsum, s), "grad_scale launch");
#ifdef LIST_BWD_REEVAL_FC2
  // (until round 3, kept for A/B builds: the forward's fused fc_2 + fc_out epilogue wrote no H3, so fc_2 ran again here)
  GemmParams gp;
  memset(&gp, 0, sizeof(gp));
  gp.fmt = fmt;
  gp.a_hi = fw + ws.h2_hi; gp.a_lo = fw + ws.h2_lo;
  gp.w_hi = wp + pk.w2_hi; gp.w_lo = wp + pk.w2_lo;
  gp.bias = (const float*)(wp + pk.b2);
  gp.M = crow; gp.N = a->H3; gp.K = a->H2;
  gp.out_hi = plane(bw.h3_hi); gp.out_lo = lo ? plane(bw.h3_lo) : nullptr; gp.ldo = a->H3;
  LIST_TRY(launch_gemm(gp, terms, EPI_RELU_SPLIT, s), "fc_2 re-evaluation launch");
  unsigned short* const h3_hi = plane(bw.h3_hi);
  unsigned short* const h3_lo = lo ? plane(bw.h3_lo) : nullptr;
#else
  unsigned short* const h3_hi = (unsigned short*)(fw + ws.h3_hi);
  unsigned short* const h3_lo = lo ? (unsigned short*)(fw + ws.h3_lo) : nullptr;
#endif
  LIST_TRY(launch_head(ga->grad_sdf, order, n_valid, crow, a->H3, h3_hi, (const float*)(wp + pk.w3),
                       scale, plane(bw.dz3_hi), lo ? plane(bw.dz3_lo) : nullptr, fmt, s), "head launch");
  if (ga->mlp.w3)
    LIST_TRY(side_colsum(h3_hi, h3_lo, a->H3, ga->grad_sdf, order, 0, ga->mlp.w3), "d fc_out.weight launch");
  if (ga->mlp.b2)
    LIST_TRY(side_colsum(plane(bw.dz3_hi), lo ? plane(bw.dz3_lo) : nullptr, a->H3, nullptr, nullptr, 1, ga->mlp.b2),
             "d fc_2.bias launch");
  mark(LIST_BWD_HEAD);

  // one layer of the chain: weight gradient (TN), bias gradient, then the masked data gradient (NT)
  auto wgrad = [&](size_t dz_hi, size_t dz_lo, int M, const char* act_hi, const char* act_lo, int N, int ldb,
                   const FeatLayout* layout, float* out, int ldo, hipStream_t s) -> hipError_t {
    if (!out) return hipSuccess;
    GemmTnParams tp;
    memset(&tp, 0, sizeof(tp));
    tp.a_hi = bwp + dz_hi; tp.a_lo = lo ? bwp + dz_lo : nullptr; tp.lda = M;
    tp.b_hi = act_hi; tp.b_lo = lo ? act_lo : nullptr; tp.ldb = ldb;
    tp.M = M; tp.N = N; tp.P = crow;
    tp.splits = wgrad_splits(M, N, crow, terms);
    const int nk = crow / (terms == 3 ? 32 : 64);
    tp.steps_per_split = (nk + tp.splits - 1) / tp.splits;
    tp.slab = slab; tp.ldn = N; tp.fmt = fmt;
    tp.b_x3i = (layout && !fp16) ? 1 : 0;          // B = X: hi / lo halfs interleaved in the split formats
    hipError_t err = launch_gemm_tn(tp, terms, s);
    if (err != hipSuccess) return err;
    return launch_wgrad_reduce(slab, tp.splits, M, N, N, layout, scale, out, ldo, s);
  };
  auto dgrad = [&](size_t dz_hi, size_t dz_lo, int K, const char* wt_hi, const char* wt_lo, int N,
                   const char* mask_hi, size_t out_hi, size_t out_lo) -> hipError_t {
    GemmParams d;
    memset(&d, 0, sizeof(d));
    d.fmt = fmt;
    d.a_hi = bwp + dz_hi; d.a_lo = bwp + dz_lo;
    d.w_hi = wt_hi; d.w_lo = wt_lo;
    d.M = crow; d.N = N; d.K = K;
    d.out_hi = plane(out_hi); d.out_lo = lo ? plane(out_lo) : nullptr; d.ldo = N;
    d.mask = (const unsigned short*)mask_hi; d.ldmask = N;
    return launch_gemm(d, terms, EPI_MASK_SPLIT, s);
  };

  // fc_2
  // (forked: the weight gradients feed nothing downstream either -- they queue up on the window stream, in the order
  // dW2, dW1, dW0 because they share the split-K slab, while the data-gradient chain continues on `s`)
  if (forked && ga->mlp.w2) LIST_TRY(hand_over(s, s_window), "stream fork");
  LIST_TRY(wgrad(bw.dz3_hi, bw.dz3_lo, a->H3, fw + ws.h2_hi, fw + ws.h2_lo, a->H2, a->H2, nullptr, ga->mlp.w2,
                 a->H2, s_window), "dW2 launch");
  mark(LIST_BWD_WGRAD2);
  LIST_TRY(dgrad(bw.dz3_hi, bw.dz3_lo, a->H3, wt + pb.w2t_hi, wt + pb.w2t_lo, a->H2, fw + ws.h2_hi, bw.dz2_hi,
                 bw.dz2_lo), "dH2 launch");
  mark(LIST_BWD_DGRAD2);
  // fc_1
  if (ga->mlp.b1)
    LIST_TRY(side_colsum(plane(bw.dz2_hi), lo ? plane(bw.dz2_lo) : nullptr, a->H2, nullptr, nullptr, 1, ga->mlp.b1),
             "d fc_1.bias launch");
  if (forked && ga->mlp.w1) LIST_TRY(hand_over(s, s_window), "stream fork");
  LIST_TRY(wgrad(bw.dz2_hi, bw.dz2_lo, a->H2, fw + ws.h1_hi, fw + ws.h1_lo, a->H1, a->H1, nullptr, ga->mlp.w1,
                 a->H1, s_window), "dW1 launch");
  mark(LIST_BWD_WGRAD1);
  LIST_TRY(dgrad(bw.dz2_hi, bw.dz2_lo, a->H2, wt + pb.w1t_hi, wt + pb.w1t_lo, a->H1, fw + ws.h1_hi, bw.dz1_hi,
                 bw.dz1_lo), "dH1 launch");
  mark(LIST_BWD_DGRAD1);
  // fc_0
  if (ga->mlp.b0)
    LIST_TRY(side_colsum(plane(bw.dz1_hi), lo ? plane(bw.dz1_lo) : nullptr, a->H1, nullptr, nullptr, 1, ga->mlp.b0),
             "d fc_0.bias launch");
  bool want_maps = ga->grad_img_map || ga->grad_trans_mat || ga->grad_percep_feat;
  for (int l = 0; l < LIST_N_VOX_LEVELS; ++l) want_maps = want_maps || ga->grad_vox[l].data;
  if (!want_maps) {
    LIST_TRY(hand_over(s_window, s), "stream join");            // dW0 shares the slab with dW2 / dW1
    LIST_TRY(wgrad(bw.dz1_hi, bw.dz1_lo, a->H1, fw + ws.x_hi, fw + ws.x_lo, L.Kp, L.Kp, &L, ga->mlp.w0, L.F, s),
             "dW0 launch");
    for (int st = LIST_BWD_DGRAD0; st < LIST_N_BWD_STAGES; ++st) mark(st);
    LIST_TRY(hand_over(s_direct, s), "stream join");
    return LIST_OK;
  }
  {
    GemmParams d;
    memset(&d, 0, sizeof(d));
    d.fmt = fmt;
    d.a_hi = bwp + bw.dz1_hi; d.a_lo = bwp + bw.dz1_lo;
    d.w_hi = wt + pb.w0t_hi; d.w_lo = wt + pb.w0t_lo;
    d.M = crow; d.N = pb.KpT; d.K = a->H1;
    d.dx = bwp + bw.dx; d.dx_f16 = fp16 ? 1 : 0; d.n_store = L.Kp; d.ldo = L.Kp;
    LIST_TRY(launch_gemm(d, terms, EPI_DX, s), "dX launch");
  }
  mark(LIST_BWD_DGRAD0);

  // dX is ready on `s`.  From here the work is a DAG of stages bound by different units: dW0 (MFMA; needs only
  // dZ1 and X) and the 16^3 LDS-window level on one stream, the atomic-rate-bound levels on another, the 8^3 window level
  // on the third (aux_streams[2]; without it, behind the gathers), the gathers (voxel-side gather, perceptual map,
  // trans_mat) on `s`.  Without auxiliary streams: the same order, in line.
  LIST_TRY(hand_over(s, s_direct), "stream fork");
  LIST_TRY(hand_over(s, s_window), "stream fork");
  LIST_TRY(hand_over(s, s_win2), "stream fork");
  if (!(bwd_knockout() & 1))
  LIST_TRY(wgrad(bw.dz1_hi, bw.dz1_lo, a->H1, fw + ws.x_hi, fw + ws.x_lo, L.Kp, L.Kp, &L, ga->mlp.w0, L.F, s_window),
           "dW0 launch");
  mark(LIST_BWD_WGRAD0);
  if (forked && ga->grad_trans_mat && !a->percep_feat) {        // (see the trans_mat gradient below)
    LIST_TRY(hipEventCreateWithFlags(&ev_dw0, hipEventDisableTiming), "event");
    LIST_TRY(hipEventRecord(ev_dw0, s_window), "event");
  }
  LIST_TRY(launch_scatter_vox(sp, L, *a, ga->grad_vox, vb, sst), "voxel scatter launch");
  mark(LIST_BWD_VOX);
  if (a->percep_feat) {
    if (ga->grad_percep_feat)
      LIST_TRY(launch_rows_to_grad(sp, L.img_off, L.img_C, a->B, (int*)(bwp + bw.vs_keys), ga->grad_percep_feat,
                                   ga->gpf_sb, ga->gpf_sc, ga->gpf_sn, s), "grad_percep_feat launch");
    mark(LIST_BWD_IMG); mark(LIST_BWD_TRANS);
    LIST_TRY(hand_over(s_direct, s), "stream join");
    LIST_TRY(hand_over(s_window, s), "stream join");
    LIST_TRY(hand_over(s_win2, s), "stream join");
    return LIST_OK;
  }
  const int nslots = a->B < kSortImages ? a->B : kSortImages;
  const int* bins_pix = pix ? (const int*)(fw + ws.bins) + (size_t)nslots * kSortCells : nullptr;
  if (pix && a->B > kSortImages) { sp.g.order_img = nullptr; sp.g.row_of = nullptr; bins_pix = nullptr; }
  const int map_f16 = ga->grad_img_map_dtype == LIST_MAP_F16 ? 1 : 0;
  // Forked, k_trans_grad must not run while dW0 does.  Whenever one of its waves shared a SIMD with the weight-gradient
  // GEMM's (2 x 216 + 80 registers = the whole file in the split formats), ONE point's v-derivative came out different --
  // d_trans_mat off by 1e-4 .. 7e-3 of its largest entry in 4 - 59 of 60 calls, depending on what else ran (tools/
  // trans_noise_probe2.py, profiles/r04b_trans_mat_interference.txt: never with dW0 left out, never in line, never once
  // the kernel held more registers or LDS than fit beside dW0; its inputs, its LDS records and every other gradient
  // were bit-stable).  It was later localised to the one ds_bpermute pair of the kernel's loop, which is gone
  // (bwd_scatter_kernels.hip); the order stays as a second fence and removes the overlap: map gradient, adjoint
  // resize (it needs only the map gradient), then -- behind dW0's event -- the trans_mat gradient.  In line the stages
  // keep their order (and their stage events their meaning).
  if (!(bwd_knockout() & 32)) {
    // (LIST_BWD_TRANS_UNORDERED=1, diagnostic: the old order -- the stage right behind the map gradient, beside dW0 --
    // for tools/trans_noise_probe2.py)
    static const bool unordered = [] { const char* e = getenv("LIST_BWD_TRANS_UNORDERED"); return e && e[0] == '1'; }();
    const bool split = forked && ga->grad_trans_mat && !unordered;
    LIST_TRY(launch_img_grad(sp, L, *a, bins_pix, nslots, bwp + bw.recs, ga->grad_img_map, map_f16,
                             split ? nullptr : ga->grad_trans_mat, ga->stage_events, s, bwp + bw.img_heavy,
                             bw.img_heavy_bytes), "image gradient launch");
    if (ga->grad_img_levels)
      LIST_TRY(launch_img_grad_to_levels(ga->grad_img_map, a->B, a->map_size, L.img_C, ga->grad_img_levels, s, map_f16,
                                         scale), "img_grad_to_levels launch");
    if (split) {
      // ... on aux_streams[2] where there is one (idle once the 8^3 window level is done) instead of at the end of `s`:
      // training step 6.43 -> 6.36 ms, 6.52 -> 6.32 with the points on the clamp (LIST_BWD_TRANS_WIN2=0: on `s`)
      static const bool on_win2 = [] { const char* e = getenv("LIST_BWD_TRANS_WIN2"); return !(e && e[0] == '0'); }();
      const hipStream_t st = (on_win2 && s_win2 != s) ? s_win2 : s;
      if (ev_dw0) LIST_TRY(hipStreamWaitEvent(st, ev_dw0, 0), "stream order");
      LIST_TRY(launch_img_grad(sp, L, *a, bins_pix, nslots, bwp + bw.recs, nullptr, map_f16, ga->grad_trans_mat,
                               ga->stage_events, st, nullptr, 0), "trans_mat gradient launch");
    }
  }
  if (bwd_knockout() & 32) { mark(LIST_BWD_IMG); mark(LIST_BWD_TRANS); }
  if (ev_dw0) { (void)hipEventDestroy(ev_dw0); ev_dw0 = nullptr; }
  LIST_TRY(hand_over(s_direct, s), "stream join");
  LIST_TRY(hand_over(s_window, s), "stream join");
  LIST_TRY(hand_over(s_win2, s), "stream join");
#undef LIST_TRY
  return LIST_OK;
}

size_t list_percep_pool_bwd_workspace_bytes(int64_t n_points, int32_t img_C) {
  if (n_points <= 0 || img_C <= 0) return 0;
  const int64_t rows = (n_points + kGatherRows - 1) / kGatherRows * kGatherRows;
  return align_up((size_t)rows * img_C * 4, 256) + align_up((size_t)rows * 32, 256) + 256;
}

int list_percep_pool_bwd(const ListPoolGradArgs* ga, void* stream) {
  if (!ga || !ga->fwd) return fail(LIST_ERR_ARG, "args/fwd is NULL");
  const ListPoolArgs* a = ga->fwd;
  if (!a->pc || !a->trans_mat || !a->img_map || !ga->grad_out || !ga->workspace) return fail(LIST_ERR_ARG, "NULL pointer");
  if (a->B <= 0 || a->N <= 0 || a->map_size < 2 || a->img_C <= 0 || a->img_C % 4)
    return fail(LIST_ERR_SHAPE, "B=%d N=%d map_size=%d img_C=%d", a->B, a->N, a->map_size, a->img_C);
  if (!dtype_ok(a->img_dtype) || (a->img_dtype == LIST_MAP_F16 && a->img_C % 8))
    return fail(LIST_ERR_ARG, "img_dtype=%d img_C=%d", a->img_dtype, a->img_C);
  const int64_t P = (int64_t)a->B * a->N;
  if (P * a->img_C >= (int64_t)1 << 31) return fail(LIST_ERR_SHAPE, "B*N*img_C must stay below 2^31");
  if (ga->workspace_bytes < list_percep_pool_bwd_workspace_bytes(P, a->img_C))
    return fail(LIST_ERR_WORKSPACE, "workspace too small: %zu < %zu", ga->workspace_bytes,
                list_percep_pool_bwd_workspace_bytes(P, a->img_C));
  if (!aligned16(ga->workspace) || !aligned16(a->img_map)) return fail(LIST_ERR_SHAPE, "workspace/img_map must be 16-byte aligned");
  const int64_t rows = (P + kGatherRows - 1) / kGatherRows * kGatherRows;
  char* wsb = (char*)ga->workspace;
  float* dx = (float*)wsb;
  char* recs = wsb + align_up((size_t)rows * a->img_C * 4, 256);
  float* scale = (float*)(recs + align_up((size_t)rows * 32, 256));
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = launch_grad_to_rows(ga->grad_out, ga->g_sb, ga->g_sc, ga->g_sn, a->B, a->N, a->img_C, dx, scale, s);
  if (e != hipSuccess) return hip_fail(e, "grad_to_rows launch");
  // the shared kernels take the fused call's descriptors: a point set without orders, X = the gradient rows
  ListQueryArgs q;
  memset(&q, 0, sizeof(q));
  q.B = a->B; q.N = a->N; q.trans_mat = a->trans_mat; q.img_map = a->img_map; q.img_dtype = a->img_dtype;
  q.map_size = a->map_size; q.img_C = a->img_C; q.clamp_hi = a->clamp_hi;
  FeatLayout L;
  memset(&L, 0, sizeof(L));
  L.img_off = 0; L.img_C = a->img_C; L.Kp = a->img_C;
  ScatterParams sp;
  memset(&sp, 0, sizeof(sp));
  sp.g.query = a->pc; sp.g.q_sb = a->p_sb; sp.g.q_sn = a->p_sn; sp.g.q_sc = a->p_sc;
  sp.g.perm0 = 0; sp.g.perm1 = 1; sp.g.perm2 = 2; sp.g.scale = 1.f;
  sp.g.N = a->N; sp.g.p_begin = 0; sp.g.n_valid = (int)P; sp.g.rows = (int)rows; sp.g.Kp = a->img_C;
  sp.dx = dx; sp.dx_f16 = 0; sp.scale = scale;
  e = launch_img_grad(sp, L, q, nullptr, 0, recs, ga->grad_img_map, 0, ga->grad_trans_mat, nullptr, s);
  if (e != hipSuccess) return hip_fail(e, "percep_pool_bwd launch");
  return LIST_OK;
}

int list_img_map_grad_to_levels(const float* grad_img_map, int32_t B, int32_t map_size,
                                const ListMap2D grads[LIST_N_IMG_LEVELS], void* stream) {
  if (!grad_img_map || !grads) return fail(LIST_ERR_ARG, "NULL pointer");
  if (B <= 0 || map_size < 2 || map_size > 320)
    return fail(LIST_ERR_SHAPE, "B=%d map_size=%d (need 2..320)", B, map_size);
  int Ct = 0;
  for (int i = 0; i < LIST_N_IMG_LEVELS; ++i) {
    if (grads[i].C < 1 || grads[i].H < 1 || grads[i].W < 1)
      return fail(LIST_ERR_SHAPE, "image level %d: bad descriptor", i);
    Ct += grads[i].C;
  }
  hipError_t e = launch_img_grad_to_levels(grad_img_map, B, map_size, Ct, grads, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "img_grad_to_levels launch");
  return LIST_OK;
}

int list_gemm_tn(const void* a_hi, const void* a_lo, const void* b_hi, const void* b_lo, float* out,
                 void* slab, size_t slab_bytes, int32_t M, int32_t N, int32_t P, int32_t precision,
                 void* stream) {
  if (!a_hi || !b_hi || !out || !slab) return fail(LIST_ERR_ARG, "NULL pointer");
  if (precision < LIST_PREC_BF16X3 || precision > LIST_PREC_FP16) return fail(LIST_ERR_ARG, "precision=%d", precision);
  if (precision == LIST_PREC_BF16X3 && (!a_lo || !b_lo)) return fail(LIST_ERR_ARG, "lo planes missing");
  if (M <= 0 || M % 256 || N < 8 || N % 8 || P <= 0 || P % 256)
    return fail(LIST_ERR_SHAPE, "M=%d N=%d P=%d (need M,P %% 256 == 0, N %% 8 == 0)", M, N, P);
  if (!aligned16(a_hi) || !aligned16(b_hi) || (a_lo && !aligned16(a_lo)) || (b_lo && !aligned16(b_lo)))
    return fail(LIST_ERR_SHAPE, "operands must be 16-byte aligned");
  const int terms = precision == LIST_PREC_BF16X3 ? 3 : 1;
  GemmTnParams tp;
  memset(&tp, 0, sizeof(tp));
  tp.a_hi = (const char*)a_hi; tp.a_lo = (const char*)a_lo; tp.lda = M;
  tp.b_hi = (const char*)b_hi; tp.b_lo = (const char*)b_lo; tp.ldb = N;
  tp.M = M; tp.N = N; tp.P = P;
  tp.splits = wgrad_splits(M, N, P, terms);
  const int nk = P / (terms == 3 ? 32 : 64);
  tp.steps_per_split = (nk + tp.splits - 1) / tp.splits;
  tp.slab = (float*)slab; tp.ldn = N; tp.fmt = precision == LIST_PREC_FP16 ? FMT_FP16 : FMT_BF16_SPLIT;
  if (slab_bytes < (size_t)tp.splits * M * N * 4)
    return fail(LIST_ERR_WORKSPACE, "slab too small: %zu < %zu", slab_bytes, (size_t)tp.splits * M * N * 4);
  hipError_t e = launch_gemm_tn(tp, terms, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "gemm_tn launch");
  e = launch_wgrad_reduce(tp.slab, tp.splits, M, N, N, nullptr, nullptr, out, N, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "wgrad_reduce launch");
  return LIST_OK;
}

// ------------------------------------------------------------------------------------------ diagnostics
int list_gemm_nt(const void* a_hi, const void* a_lo, const void* w_hi, const void* w_lo,
                 const float* bias, float* out, int32_t M, int32_t N, int32_t K, int32_t relu,
                 int32_t precision, void* stream) {
  if (!a_hi || !w_hi || !out) return fail(LIST_ERR_ARG, "NULL pointer");
  if (precision == LIST_PREC_BF16X3 && (!a_lo || !w_lo)) return fail(LIST_ERR_ARG, "lo planes missing");
  if (precision < LIST_PREC_BF16X3 || precision > LIST_PREC_FP16) return fail(LIST_ERR_ARG, "precision=%d", precision);
  if (M <= 0 || M % 256 || N <= 0 || N % 256 || K <= 0 || K % 64)
    return fail(LIST_ERR_SHAPE, "M=%d N=%d K=%d (need M,N %% 256 == 0, K %% 64 == 0)", M, N, K);
  if (!aligned16(a_hi) || !aligned16(w_hi) || (a_lo && !aligned16(a_lo)) || (w_lo && !aligned16(w_lo)))
    return fail(LIST_ERR_SHAPE, "operands must be 16-byte aligned");
  GemmParams gp;
  memset(&gp, 0, sizeof(gp));
  gp.a_hi = (const char*)a_hi; gp.a_lo = (const char*)a_lo;
  gp.w_hi = (const char*)w_hi; gp.w_lo = (const char*)w_lo;
  gp.bias = bias; gp.M = M; gp.N = N; gp.K = K; gp.out_f32 = out; gp.relu = relu & 1;
  gp.plain_loop = (relu & 2) ? 1 : 0;
  gp.x3i = (relu & 4) ? 1 : 0;                     // a_hi / w_hi hold hi and lo interleaved (bf16 formats only)
  if (gp.x3i && precision == LIST_PREC_FP16) return fail(LIST_ERR_ARG, "interleaved operands are a bf16-format layout");
  gp.fmt = precision == LIST_PREC_FP16 ? FMT_FP16 : FMT_BF16_SPLIT;
  hipError_t e = launch_gemm(gp, precision == LIST_PREC_BF16X3 ? 3 : 1, EPI_F32, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "gemm launch");
  return LIST_OK;
}

int list_split_bf16(const float* x, void* hi, void* lo, int64_t n, void* stream) {
  if (!x || !hi) return fail(LIST_ERR_ARG, "NULL pointer");
  if (n <= 0 || n % 4) return fail(LIST_ERR_SHAPE, "n=%lld must be a positive multiple of 4", (long long)n);
  if (!aligned16(x)) return fail(LIST_ERR_SHAPE, "x must be 16-byte aligned");
  hipError_t e = launch_split(x, (unsigned short*)hi, (unsigned short*)lo, n, FMT_BF16_SPLIT,
                              (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "split launch");
  return LIST_OK;
}

int list_to_fp16(const float* x, void* out, int64_t n, void* stream) {
  if (!x || !out) return fail(LIST_ERR_ARG, "NULL pointer");
  if (n <= 0 || n % 4) return fail(LIST_ERR_SHAPE, "n=%lld must be a positive multiple of 4", (long long)n);
  if (!aligned16(x)) return fail(LIST_ERR_SHAPE, "x must be 16-byte aligned");
  hipError_t e = launch_split(x, (unsigned short*)out, nullptr, n, FMT_FP16, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "to_fp16 launch");
  return LIST_OK;
}

}  // extern "C"
