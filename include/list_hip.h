/*
 * list_hip.h -- C ABI of the MI355X (gfx950) implementation of LIST's SDF query hot path.
 *
 * The reference (robotic-vision-lab/Learning-Implicitly-From-Spatial-Transformers-Network)
 * has no FFI layer: its boundary for this path is the Python nn.Module API.  Each entry point
 * below replaces one group of PyTorch op call sites of the reference and is what a binding
 * (ctypes / torch extension) for that path calls; INTEGRATION.md shows the binding.
 *
 * Conventions (all entry points)
 *   - plain C, POD structs, raw DEVICE pointers, explicit shapes and ELEMENT strides;
 *   - the caller owns every buffer (inputs, outputs, workspace): nothing is allocated,
 *     freed or retained by the library, no hipMalloc/hipFree/sync inside (graph-capture safe);
 *   - work is enqueued on the caller's stream (hipStream_t passed as void*), asynchronously;
 *   - the device is the caller's current device (hipSetDevice before the call);
 *   - return LIST_OK (0) or a negative error code; never throws, never aborts;
 *     list_last_error() returns a thread-local description of the last failure;
 *   - re-entrant and thread-safe: no global mutable state besides the thread-local
 *     error string.
 */
#ifndef LIST_HIP_H
#define LIST_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LIST_ABI_VERSION 9

#define LIST_N_IMG_LEVELS 5   /* ResEncoder feature maps, network/modules.py:1067 */
#define LIST_N_VOX_LEVELS 6   /* VoxelEncoder2 feature maps, network/modules.py:425-442 */
#define LIST_N_STENCIL 7      /* displacement stencil, network/modules.py:205-214 */

enum ListStatus {
  LIST_OK = 0,
  LIST_ERR_ARG = -1,         /* null pointer / bad enum / bad struct */
  LIST_ERR_SHAPE = -2,       /* shape or stride the kernels do not support */
  LIST_ERR_WORKSPACE = -3,   /* workspace or output buffer too small */
  LIST_ERR_HIP = -4,         /* a HIP launch failed; see list_last_error() */
  LIST_ERR_UNSUPPORTED = -5
};

/* Arithmetic of the implicit MLP (the gathers and all interpolation are always fp32; the
 * accumulation is always fp32):
 *   BF16X3 : operands split into bf16 hi+lo, 3 MFMA products per MAC (measured |err| < 1e-6 of
 *            the fp32 reference: fp32-grade);
 *   BF16   : bf16 operands, 1 MFMA per MAC (measured |err| ~ 3e-4, stated tolerance 5e-3);
 *   FP16   : fp16 operands (11-bit significand), 1 MFMA per MAC (measured |err| ~ 4e-5, inside
 *            the 1e-4 parity bound); magnitudes above 65504 saturate.
 * The packed weights (list_prep_mlp_weights) are specific to {BF16X3,BF16} or {FP16}. */
enum ListPrecision { LIST_PREC_BF16X3 = 0, LIST_PREC_BF16 = 1, LIST_PREC_FP16 = 2 };

/* Element type of the PREPARED (channels-last) maps the gathers read.  F16 halves the bytes of the
 * layout hand-off and of every tap (round-to-nearest-even, saturating at +-65504); interpolation is
 * fp32 either way.  Meant to be paired with LIST_PREC_FP16, whose features are rounded to fp16 after
 * interpolation anyway (measured |err| of the pair ~5e-5, inside the 1e-4 bound).  Needs every vector
 * voxel level and the perceptual map to have a channel count that is a multiple of 8. */
enum ListMapDtype { LIST_MAP_F32 = 0, LIST_MAP_F16 = 1 };

/* One 2-D feature map [B,C,H,W] float32 with element strides (NCHW or channels-last). */
typedef struct ListMap2D {
  const float* data;
  int32_t C, H, W;
  int64_t sb, sc, sh, sw;
} ListMap2D;

/* One 3-D feature map [B,C,D,H,W] with element strides.  dtype (enum ListMapDtype): LIST_MAP_F32, or
 * LIST_MAP_F16 for a producer that runs in half precision (a channels-last fp16 level with
 * C % 8 == 0 is then used where it lies when fp16 maps are asked for; anything else is converted). */
typedef struct ListMap3D {
  const void* data;
  int32_t C, D, H, W;
  int32_t dtype, reserved_;
  int64_t sb, sc, sd, sh, sw;
} ListMap3D;

/* A voxel level in the layout the gather kernels read: per image [D][H][W][C] contiguous. */
typedef struct ListVoxLevel {
  const void* data;          /* float or half elements, see dtype */
  int32_t C, D, H, W;
  int32_t dtype;             /* enum ListMapDtype */
  int32_t reserved_;
  int64_t image_stride;      /* elements between images */
} ListVoxLevel;

/* MLP parameters in the REFERENCE layout (state_dict of VoxelDecoder.fc,
 * network/modules.py:196-200; Conv1d weights [out,in,1] are [out,in] row-major). */
typedef struct ListMlpWeights {
  const float* w0; const float* b0;   /* fc_0  : [H1, F],  [H1]   F = 7*sum(C_vox)+C_img+3 */
  const float* w1; const float* b1;   /* fc_1  : [H2, H1], [H2] */
  const float* w2; const float* b2;   /* fc_2  : [H3, H2], [H3] */
  const float* w3; const float* b3;   /* fc_out: [1, H3],  [1]  */
  int32_t F, H1, H2, H3;
  int32_t vox_C[LIST_N_VOX_LEVELS];   /* channel count per voxel level (feature order k=c*7+j) */
  int32_t img_C;                      /* total perceptual channels (1024) */
  int32_t precision;                  /* enum ListPrecision the packed copy is built for */
} ListMlpWeights;

/* ---------------------------------------------------------------------------------------
 * list_prep_img_maps -- replaces F.interpolate(x_i, 137, bilinear, align_corners=True) x5
 * (network/modules.py:26-35) and fixes the layout: writes ONE channels-last map
 * out[B][map_size][map_size][sum C_i] (channel order = concatenation order of modules.py:53).
 * Returns LIST_OK or an error.  Required out size: list_img_map_bytes().
 */
size_t list_img_map_bytes(const ListMap2D maps[LIST_N_IMG_LEVELS], int32_t B, int32_t map_size,
                          int32_t map_dtype);
int list_prep_img_maps(const ListMap2D maps[LIST_N_IMG_LEVELS], int32_t B, int32_t map_size,
                       int32_t map_dtype, void* out, size_t out_bytes, void* stream);

/* ---------------------------------------------------------------------------------------
 * list_prep_vox_maps -- layout hand-off for F.grid_sample 3-D (network/modules.py:263-265):
 * converts each [B,C,D,H,W] map to per-image [D][H][W][C].  A level that already has that
 * layout (channels_last_3d strides, or C == 1 with contiguous D,H,W) is used in place (as fp32)
 * and costs nothing; converted levels are written in `map_dtype`.  levels_out[] (host memory)
 * receives the descriptors for list_sdf_query_fwd.
 */
size_t list_vox_pack_bytes(const ListMap3D maps[LIST_N_VOX_LEVELS], int32_t B, int32_t map_dtype);
int list_prep_vox_maps(const ListMap3D maps[LIST_N_VOX_LEVELS], int32_t B, int32_t map_dtype,
                       void* pack, size_t pack_bytes, ListVoxLevel levels_out[LIST_N_VOX_LEVELS],
                       void* stream);

/* ---------------------------------------------------------------------------------------
 * list_prep_mlp_weights -- one-off repack of the Conv1d(k=1) parameters
 * (network/modules.py:196-200): fc_0 columns permuted from the reference feature order
 * (k = c*7+j | perceptual | xyz, modules.py:270-275) to the gather order, K padded,
 * bf16 hi/lo split.  Call again whenever the parameters change.
 */
size_t list_packed_mlp_bytes(const ListMlpWeights* w);
int list_prep_mlp_weights(const ListMlpWeights* w, void* packed, size_t packed_bytes, void* stream);

/* ---------------------------------------------------------------------------------------
 * list_prep_percep_proj -- inference with many query points per image (a marching-cubes grid,
 * network/executors.py:191-231): fc_0 is linear, so the contribution of the perceptual block of
 * the feature vector (modules.py:46-53: bilinear samples of the prepared map) to fc_0's
 * pre-activation equals the bilinear sample of the PROJECTED map
 *   proj[b][y][x][n] = sum_c img_map[b][y][x][c] * fc_0.weight[n][perceptual column c]
 * -- one [B*map_size^2, img_C] x [img_C, H1] product per image instead of img_C columns of fc_0 per
 * query point.  Same arithmetic class as the MLP (operand format of `precision`, fp32 accumulate);
 * the projected map is fp16 for LIST_PREC_FP16 (img_map must be LIST_MAP_F16) and fp32 for the bf16
 * formats (img_map must be LIST_MAP_F32; `scratch` then holds its bf16 hi/lo copy).  Valid while
 * img_map and packed_mlp stay unchanged.  Finite maps: the exact border / non-finite semantics of the
 * standard path are per feature and are not reproduced tap by tap here.
 */
size_t list_percep_proj_bytes(int32_t B, int32_t map_size, int32_t H1, int32_t precision);
size_t list_percep_proj_scratch_bytes(int32_t B, int32_t map_size, int32_t img_C, int32_t precision);
int list_prep_percep_proj(const void* img_map, int32_t img_dtype, int32_t B, int32_t map_size,
                          const int32_t vox_C[LIST_N_VOX_LEVELS], int32_t img_C, const void* packed_mlp,
                          int32_t H1, int32_t H2, int32_t H3, int32_t precision, void* proj,
                          size_t proj_bytes, void* scratch, size_t scratch_bytes, void* stream);

/* ---------------------------------------------------------------------------------------
 * list_prep_img_proj (ABI 8) -- inference forwards at ANY number of query points: the same linearity, applied
 * BEFORE the resize.  F.interpolate (network/modules.py:26-35) is linear per channel with the same weights for every
 * channel, so for an encoder level l
 *   sum_c fc_0.weight[n][c of level l] * resize(x_l)[b][c][y][x]  ==  resize(P_l)[b][n][y][x],
 *   P_l[b][n][v][u] = sum_c fc_0.weight[n][c of level l] * x_l[b][c][v][u]       (H_l x W_l source pixels),
 * and the projection costs H_l * W_l * C_l * H1 MACs per image instead of C_l * H1 per query point: 14^2 ... 56^2
 * pixels against 20 000 points for the three low-resolution levels, which hold 896 of the 1024 perceptual channels.
 * Levels 0 .. n_kept_levels-1 (the high-resolution ones, few channels) are resized as list_prep_img_maps does; the
 * levels from n_kept_levels on are projected at their own resolution, resized and SUMMED.  Writes ONE channels-last
 * map  out[B][map_size][map_size][kept_C + H1]  (kept_C = channels of the kept levels; fp16 elements for
 * LIST_PREC_FP16, fp32 for the bf16 formats) for ListQueryArgs.img_proj = 1: per point the kept channels are sampled
 * into the feature matrix as before, the H1 projected channels into an fp32 row vector that fc_0 adds in its
 * epilogue, and fc_0's K loop leaves the projected levels' columns out.  Same arithmetic class as the MLP (operand
 * format of `precision`, fp32 accumulate; P_l and the resized sum are fp32).  Valid while the maps and packed_mlp stay
 * unchanged.  Finite maps (as list_prep_percep_proj).  kept_C % 64 == 0, every projected level's C % 64 == 0.
 */
size_t list_img_proj_map_bytes(const ListMap2D maps[LIST_N_IMG_LEVELS], int32_t B, int32_t map_size,
                               int32_t n_kept_levels, int32_t H1, int32_t precision);
size_t list_img_proj_scratch_bytes(const ListMap2D maps[LIST_N_IMG_LEVELS], int32_t B, int32_t n_kept_levels,
                                   int32_t H1, int32_t precision);
int list_prep_img_proj(const ListMap2D maps[LIST_N_IMG_LEVELS], int32_t B, int32_t map_size,
                       int32_t n_kept_levels, const int32_t vox_C[LIST_N_VOX_LEVELS], const void* packed_mlp,
                       int32_t H1, int32_t H2, int32_t H3, int32_t precision, void* out, size_t out_bytes,
                       void* scratch, size_t scratch_bytes, void* stream);

/* ---------------------------------------------------------------------------------------
 * list_sdf_query_fwd -- the fused hot path: LIST.forward lines network/models.py:91-97, i.e.
 * PerceptualPooling.forward (modules.py:37-53, on the prepared 137^2 map) +
 * VoxelDecoder2.forward (modules.py:255-282).  sdf[b][n] for every query point.
 */
typedef struct ListQueryArgs {
  int32_t B, N;                         /* images, query points per image */
  const float* query;                   /* [B,N,3] */
  int64_t q_sb, q_sn, q_sc;             /* element strides of query */
  int32_t perm[3];                      /* p[i] = scale * query[..., perm[i]]; {2,1,0}, 2.0 for */
  float scale;                          /*   raw queries (models.py:91-92); {0,1,2}, 1.0 if done */
  const float* trans_mat;               /* [B,4,3] contiguous (models.py:86) */
  const void* img_map;                  /* output of list_prep_img_maps, or NULL with percep_feat */
  int32_t img_dtype;                    /* enum ListMapDtype of img_map */
  int32_t map_size;                     /* 137 */
  int32_t img_C;                        /* 1024 */
  float clamp_hi;                       /* 136.0 (hard-coded in modules.py:43); samples beyond  */
                                        /*   map_size-1 read zeros (grid_sample zeros padding)   */
  const float* percep_feat;             /* optional pre-pooled features [B,img_C,N] instead of */
  int64_t pf_sb, pf_sc, pf_sn;          /*   img_map (VoxelDecoder2.forward's 3rd argument) */
  ListVoxLevel vox[LIST_N_VOX_LEVELS];  /* from list_prep_vox_maps */
  const void* packed_mlp;               /* from list_prep_mlp_weights */
  int32_t F, H1, H2, H3;                /* must match the packed weights */
  float* sdf;                           /* [B,N] contiguous, output */
  void* workspace; size_t workspace_bytes;   /* >= list_query_workspace_bytes(B*N) */
  int32_t precision;                    /* enum ListPrecision */
  void* const* stage_events;            /* optional: stage_event_sets x LIST_N_STAGES hipEvent_t    */
                                        /*   handles (host array), recorded on the stream at the    */
                                        /*   stage boundaries below; set c belongs to row chunk c   */
  int32_t no_sort;                      /* 0: process points in Morton order (default; results are  */
                                        /*   bit-identical either way), 1: keep the caller's order  */
                                        /*   -- for queries that are spatially coherent already (a  */
                                        /*   raster grid: the sort is ~6 % of such a query's time)  */
  int32_t stage_event_sets;             /* number of event sets behind stage_events (0 counts as 1): */
                                        /*   row chunk c records into set c, chunks beyond the last  */
                                        /*   set record nothing.  list_query_chunk_rows() tells how  */
                                        /*   many chunks a call takes                                */
  const void* percep_proj;              /* optional (ABI 5, inference): output of                    */
                                        /*   list_prep_percep_proj for img_map and packed_mlp.  The  */
                                        /*   perceptual block of fc_0 is then taken from it (one     */
                                        /*   4-tap sample of H1 channels per point) and fc_0 skips   */
                                        /*   that block; img_map is still required (its descriptor   */
                                        /*   fields are checked), percep_feat must be NULL, and the  */
                                        /*   call cannot be followed by list_sdf_query_bwd           */
  int32_t no_activations;               /* ABI 6.  1: nothing of this forward is kept for          */
                                        /*   list_sdf_query_bwd (inference).  fc_1, fc_2 and fc_out  */
                                        /*   then run as ONE kernel that keeps H2 in registers (fp16 */
                                        /*   operands, H2 = H3 = 256; other cases: no effect).  The  */
                                        /*   values are those of the two-launch path bit for bit     */
                                        /*   (same products, k order and summation tree).  Such a    */
                                        /*   forward may also leave columns of the feature matrix    */
                                        /*   unwritten (ABI 7: the perceptual block is produced      */
                                        /*   inside fc_0, list_query_plan().fused_fc0).  0           */
                                        /*   (default): H1, H2, H3 are left in the workspace for the */
                                        /*   backward.  list_sdf_query_bwd refuses (LIST_ERR_ARG) a  */
                                        /*   forward that had it set.                                */
  int32_t no_fused_fc0;                 /* ABI 7.  0 (default): an inference forward (no_activations) with  */
                                        /*   fp16 or plain-bf16 operands runs fc_0 as k_fc0_fused -- the     */
                                        /*   perceptual block of its A operand (modules.py:46-53) is sampled */
                                        /*   into LDS inside the kernel, no 2-D gather kernel is launched    */
                                        /*   and those columns of the feature matrix never reach HBM.  1:    */
                                        /*   the unfused pair (k_gather_img + k_gemm_nt_pp).  Same bits      */
                                        /*   either way (tests/test_fused_fc0_gpu.py).                       */
  int32_t img_proj;                     /* ABI 8.  1: img_map is the output of list_prep_img_proj for packed_mlp: */
  int32_t img_kept_C;                   /*   img_kept_C sampled channels followed by H1 projected ones per pixel  */
                                        /*   (img_C stays the channel count of the feature layout, 1024).         */
                                        /*   Inference only (no_activations = 1), percep_feat / percep_proj NULL. */
                                        /*   0 (default): img_map is the output of list_prep_img_maps             */
} ListQueryArgs;

/* Boundaries recorded into ListQueryArgs.stage_events: one hipEvent after each kernel (group) of a
 * row chunk, so that the interval [i-1, i] of a set is the duration of kernel i for that chunk.  A query
 * above the chunk size runs ceil(B*N / list_query_chunk_rows()) chunks back to back: give one set per
 * chunk and sum the intervals over the sets for the time of the whole call.  Any handle may be NULL (that
 * boundary is not recorded).  The seven gathers of a chunk are independent of each other and are dispatched without
 * the queue barrier between them (hipExtAnyOrderLaunch behind the first one); a non-NULL handle among VOX0 .. IMG
 * asks for their individual durations and puts the barriers back, which costs the step ~2 %: leave those six NULL to
 * time the group as [SORT, TAIL]. */
enum ListStage {
  LIST_STAGE_BEGIN = 0,   /* before the first kernel */
  LIST_STAGE_SORT = 1,    /* point ordering (histogram / scan / scatter, two passes) */
  LIST_STAGE_VOX0 = 2,    /* k_gather_vox of the 1st .. 5th vector voxel level (levels with C > 1, */
  LIST_STAGE_VOX1 = 3,    /*   in level order; modules.py:256-265).  Unused slots are recorded     */
  LIST_STAGE_VOX2 = 4,    /*   back to back.                                                        */
  LIST_STAGE_VOX3 = 5,
  LIST_STAGE_VOX4 = 6,
  LIST_STAGE_IMG = 7,     /* k_gather_img: projection + bilinear sample  (modules.py:37-52) */
  LIST_STAGE_TAIL = 8,    /* k_gather_tail: C == 1 level, xyz, padding   (modules.py:257) */
  LIST_STAGE_FC0 = 9,     /* k_gemm_nt: fc_0 + ReLU                       (modules.py:276) */
  LIST_STAGE_EXACT = 10,  /* k_gather_fixup + gated fc_0: exact redo of the row tiles whose fc_0 output holds */
                          /*   a NaN (reference skip semantics for +-inf / NaN map values at border taps);   */
                          /*   both launches exit at once on finite inputs                                   */
  LIST_STAGE_FC1 = 11,    /* k_gemm_nt: fc_1 + ReLU                       (modules.py:277) */
  LIST_STAGE_FC2 = 12,    /* k_gemm_nt: fc_2 + ReLU + fc_out -> sdf       (modules.py:278-281) */
  LIST_N_STAGES = 13
};

size_t list_query_workspace_bytes(int64_t n_points, int32_t F, int32_t H1, int32_t H2, int32_t H3);
/* rows (query points) one chunk of list_sdf_query_fwd processes with a workspace of `workspace_bytes`
 * (a multiple of 256, at most 262144 and at most n_points rounded up to 256); 0 if nothing fits. */
int64_t list_query_chunk_rows(size_t workspace_bytes, int64_t n_points, int32_t F, int32_t H1, int32_t H2,
                              int32_t H3);
int list_sdf_query_fwd(const ListQueryArgs* args, void* stream);

/* list_query_plan (ABI 7) -- what list_sdf_query_fwd WOULD dispatch for `args` (validated like the call itself; no
 * launch, no HIP call): for callers that account per kernel (bench.py: which launch the fc_1 interval belongs to)
 * instead of re-deriving the library's dispatch conditions from the arguments and the environment. */
typedef struct ListQueryPlan {
  int64_t rows_per_chunk;   /* = list_query_chunk_rows() for args->workspace_bytes */
  int32_t chunks;           /* row chunks the call runs back to back */
  int32_t fused_tail;       /* 1: fc_1 + fc_2 + fc_out run as ONE launch (no kernel between LIST_STAGE_EXACT and   */
                            /*    LIST_STAGE_FC1), 0: two launches                                                   */
  int32_t fc0_k;            /* K of the fc_0 launch (feature columns incl. padding, less the perceptual block when */
                            /*    args->percep_proj is given)                                                      */
  int32_t box_levels;       /* bit l: voxel level l is gathered on the matrix cores (k_gather_vox_box: a level whose  */
                            /*    stencil stays inside one cell, 128 channels, fp16 maps and fp16 operands)           */
  int32_t fused_fc0;        /* 1: fc_0 produces the perceptual block of its A operand on chip (k_fc0_fused) and no 2-D   */
                            /*    gather kernel is launched; 0: k_gather_img writes the block into X                     */
  int32_t img_proj;         /* 1: the projected levels' columns are left out of fc_0's K loop (ListQueryArgs.img_proj)      */
} ListQueryPlan;
int list_query_plan(const ListQueryArgs* args, ListQueryPlan* plan);

/* ---------------------------------------------------------------------------------------
 * list_percep_pool_fwd -- PerceptualPooling.forward alone (network/modules.py:37-53) for
 * callers that need the reference's materialised tensor: out[B][img_C][N] (= [B,1024,1,N]).
 */
typedef struct ListPoolArgs {
  int32_t B, N;
  const float* pc; int64_t p_sb, p_sn, p_sc;   /* [B,N,3] (already permuted/scaled) */
  const float* trans_mat;                       /* [B,4,3] contiguous */
  const void* img_map; int32_t img_dtype, map_size, img_C;
  float clamp_hi;
  float* out;                                   /* [B,img_C,N] contiguous */
} ListPoolArgs;
int list_percep_pool_fwd(const ListPoolArgs* args, void* stream);

/* ---------------------------------------------------------------------------------------
 * list_gather_features_fwd -- test/diagnostic entry: the gathered feature matrix in the
 * REFERENCE order and layout, out[B][F][N] float32 (= torch.cat at modules.py:275).
 * Uses the same gather kernels as list_sdf_query_fwd (values are hi+lo of the bf16 split,
 * i.e. fp32 rounded to 16 significant bits).
 */
int list_gather_features_fwd(const ListQueryArgs* args, float* out, void* stream);

/* ---------------------------------------------------------------------------------------
 * list_gemm_nt -- test/diagnostic entry for the MFMA kernel used by the MLP:
 * out[M][N] = act(A[M][K] . W[N][K]^T + bias), A and W given as bf16 hi/lo planes
 * (lo may be NULL with LIST_PREC_BF16; with LIST_PREC_FP16 the hi planes hold fp16 and lo is
 * ignored).  M % 256 == 0, N % 256 == 0, K % 64 == 0.  relu: bit 0 = apply ReLU; bit 1 = force the plain
 * 2-stage loop instead of the ping-pong schedule that long-K single-plane products take (the two are
 * bit-identical by construction: same accumulation order -- the tests use this as a race detector);
 * bit 2 (bf16 formats) = a_hi and w_hi hold their hi and lo halfs interleaved in 64-byte blocks, the layout the
 * library keeps the feature matrix and the packed fc_0 weight in (element k of a row: hi at (k / 32) * 64 + k % 32,
 * lo 32 elements further; a_lo / w_lo are ignored) -- bit-identical to the planar operands.
 */
int list_gemm_nt(const void* a_hi, const void* a_lo, const void* w_hi, const void* w_lo,
                 const float* bias, float* out, int32_t M, int32_t N, int32_t K, int32_t relu,
                 int32_t precision, void* stream);

/* fp32 -> bf16 hi/lo planes (round-to-nearest-even; lo = bf16(x - hi)). n % 4 == 0. */
int list_split_bf16(const float* x, void* hi, void* lo, int64_t n, void* stream);
/* fp32 -> fp16 (round-to-nearest-even, saturating at +-65504). n % 4 == 0. */
int list_to_fp16(const float* x, void* out, int64_t n, void* stream);

/* =======================================================================================
 * Backward of the path (training step; the reference differentiates these ops with autograd:
 * train.py:82-85 through models.py:91-97, modules.py:24-54 and modules.py:255-282).
 *
 * list_sdf_query_bwd takes d(loss)/d(sdf) and produces the gradients of every trainable input of
 * the path: the MLP parameters (reference layout), the prepared perceptual map, the voxel levels,
 * and trans_mat.  Query coordinates are data in the reference's training loop and get no gradient.
 *
 * Contract with the forward: `fwd` are the arguments of a list_sdf_query_fwd call that has been
 * enqueued before on the same stream, with fwd->workspace UNTOUCHED since (it holds the feature
 * matrix, the hidden activations and the point order), and the whole query in ONE row chunk
 * (B*N <= 262144 and workspace_bytes >= list_query_workspace_bytes(B*N)).  With fwd->percep_feat (the
 * pre-pooled form) the perceptual part of the gradient is returned as grad_percep_feat instead of
 * grad_img_map / grad_trans_mat (list_percep_pool_bwd takes it from there).  Arithmetic follows fwd->precision: BF16X3 keeps every gradient operand as
 * bf16 hi+lo (fp32-grade), FP16 scales d(sdf) by a power of two into the fp16 range (undone in the
 * fp32 epilogues), BF16 rounds gradient operands to bf16.  All sums are fp32.
 * Outputs are OVERWRITTEN (not accumulated); a NULL output is skipped.  Map and trans_mat
 * gradients use fp32 atomics where contributions collide, so their last bits may differ run to run;
 * the MLP parameter gradients are summed in a fixed order over the rows of the feature matrix and are
 * bitwise reproducible for a fixed row order (no_sort = 1; the Morton order inside a sort bin is not
 * deterministic).
 */
typedef struct ListMlpGrads {         /* reference layouts, like ListMlpWeights */
  float* w0; float* b0;               /* [H1,F], [H1] */
  float* w1; float* b1;               /* [H2,H1], [H2] */
  float* w2; float* b2;               /* [H3,H2], [H3] */
  float* w3; float* b3;               /* [1,H3], [1] */
} ListMlpGrads;

typedef struct ListQueryGradArgs {
  const ListQueryArgs* fwd;           /* see the contract above */
  const float* grad_sdf;              /* [B,N] contiguous: d(loss)/d(sdf) */
  const void* packed_mlp_bwd;         /* from list_prep_mlp_weights_bwd (transposed copies) */
  ListMlpGrads mlp;                   /* any pointer may be NULL */
  float* grad_img_map;                /* [B][map_size][map_size][img_C] fp32 (gradient of the map    */
                                      /*   list_prep_img_maps produced), or NULL                      */
  float* grad_trans_mat;              /* [B,4,3] fp32, or NULL */
  ListVoxLevel grad_vox[LIST_N_VOX_LEVELS];   /* fp32 channels-last [B][D][H][W][C] buffers (dtype    */
                                      /*   LIST_MAP_F32, data written through); data == NULL: skipped */
  void* workspace; size_t workspace_bytes;    /* >= list_query_bwd_workspace_bytes() */
  void* const* stage_events;          /* optional: LIST_N_BWD_STAGES hipEvent_t handles */
  int32_t vox_adjoint;                /* how a voxel level's gradient is formed: 0 = chosen per level (LDS   */
                                      /*   windows for coarse levels, voxel-side gather where samples are    */
                                      /*   dense, atomics otherwise); 1 = never gather; 2 = gather wherever  */
                                      /*   the level fits the sort's bins.  Same values up to summation order */
  float* grad_percep_feat;            /* only with fwd->percep_feat (VoxelDecoder2.forward's own form): the    */
  int64_t gpf_sb, gpf_sc, gpf_sn;     /*   gradient of the pre-pooled features, [B,img_C,N] with these strides */
  void* aux_streams[3];               /* optional: more hipStream_t of the same device.  The backward is a       */
                                      /*   DAG, not a chain: dW0 (MFMA-bound), the atomic-rate-bound scatters,   */
                                      /*   the LDS-window scatters and the gathers need different units, so they */
                                      /*   are forked onto these streams (event fork/join around them: the call  */
                                      /*   is still ordered on `stream` as a whole).  [0] and [1] NULL: all in   */
                                      /*   order.  [1] carries dW0 and the 16^3 window level, the longest chain: */
                                      /*   create it with hipStreamCreateWithPriority(.., -1) (-0.06 ms per      */
                                      /*   step).  [2] (ABI 9, optional beside the other two): the 8^3 window    */
                                      /*   level, which otherwise waits behind the gathers on `stream`, and then */
                                      /*   the trans_mat gradient (-0.1 ms per step, -0.4 with the points on     */
                                      /*   the clamp)                                                            */
  const ListMap2D* grad_img_levels;   /* optional: LIST_N_IMG_LEVELS descriptors as list_img_map_grad_to_levels   */
                                      /*   takes them.  The adjoint resize then runs inside this call, beside the */
                                      /*   voxel scatters still in flight on the auxiliary streams (needs         */
                                      /*   grad_img_map as the intermediate).  NULL: call it yourself afterwards. */
  int32_t grad_img_map_dtype;         /* ABI 6.  LIST_MAP_F32 (0): grad_img_map is the fp32 gradient (above).      */
                                      /*   LIST_MAP_F16: only with grad_img_levels and fwd->precision FP16 -- the  */
                                      /*   map gradient is an INTERMEDIATE of the adjoint resize then, kept as     */
                                      /*   halfs at the gradient scale (grad_img_map holds B*ms*ms*img_C halfs of  */
                                      /*   scratch afterwards, not a gradient): half the bytes between the two     */
                                      /*   kernels (0.6 GB per step at the metric shape).  Needs the pixel-ordered */
                                      /*   gather form (point sort on, B <= 64 images, map_size*ceil(map_size/4)   */
                                      /*   <= 8192), else LIST_ERR_UNSUPPORTED.                                    */
} ListQueryGradArgs;

enum ListBwdStage {
  LIST_BWD_BEGIN = 0,
  LIST_BWD_HEAD = 1,      /* scale, fc_2 re-evaluation, d(fc_out), bias/w3 sums */
  LIST_BWD_WGRAD2 = 2,    /* k_gemm_tn: dW2 (enqueue time only when forked, like WGRAD1 / WGRAD0) */
  LIST_BWD_DGRAD2 = 3,    /* k_gemm_nt: dH2 (masked) */
  LIST_BWD_WGRAD1 = 4,
  LIST_BWD_DGRAD1 = 5,
  LIST_BWD_DGRAD0 = 6,    /* k_gemm_nt: dX [P x 3648] */
  LIST_BWD_WGRAD0 = 7,    /* k_gemm_tn: dW0 [512 x 3648], the large one (enqueue time only when forked) */
  LIST_BWD_VOX = 8,       /* scatter-add into the voxel levels */
  LIST_BWD_IMG = 9,       /* gradient of the prepared perceptual map */
  LIST_BWD_TRANS = 10,    /* gradient of trans_mat */
  LIST_N_BWD_STAGES = 11
};

size_t list_packed_mlp_bwd_bytes(const ListMlpWeights* w);
int list_prep_mlp_weights_bwd(const ListMlpWeights* w, void* packed_bwd, size_t packed_bytes, void* stream);
size_t list_query_bwd_workspace_bytes(int64_t n_points, int32_t F, int32_t H1, int32_t H2, int32_t H3,
                                      int32_t precision);
int list_sdf_query_bwd(const ListQueryGradArgs* args, void* stream);

/* Adjoint of list_prep_img_maps (the bilinear align_corners resize + concat of modules.py:26-35,53):
 * the gradient of the prepared map -> one gradient per encoder level.  grads[i].data (float32, written
 * through the given strides, every element overwritten) must describe [B,C_i,H_i,W_i]. */
int list_img_map_grad_to_levels(const float* grad_img_map, int32_t B, int32_t map_size,
                                const ListMap2D grads[LIST_N_IMG_LEVELS], void* stream);

/* ---------------------------------------------------------------------------------------
 * list_percep_pool_bwd -- backward of list_percep_pool_fwd (PerceptualPooling.forward on its own,
 * network/modules.py:37-53): d(loss)/d(out [B,img_C,N]) -> gradient of the prepared map and of trans_mat.
 * (The stand-alone form has no point sort, so the map gradient is accumulated with fp32 atomics; the fused
 * list_sdf_query_bwd is the fast path.)  Outputs are overwritten; NULL outputs are skipped.
 */
typedef struct ListPoolGradArgs {
  const ListPoolArgs* fwd;                       /* pc, trans_mat, img_map, ... of the forward call (out unused) */
  const float* grad_out; int64_t g_sb, g_sc, g_sn;   /* [B,img_C,N] with element strides */
  float* grad_img_map;                           /* [B][map_size][map_size][img_C] fp32, or NULL */
  float* grad_trans_mat;                         /* [B,4,3], or NULL */
  void* workspace; size_t workspace_bytes;       /* >= list_percep_pool_bwd_workspace_bytes(B*N, img_C) */
} ListPoolGradArgs;
size_t list_percep_pool_bwd_workspace_bytes(int64_t n_points, int32_t img_C);
int list_percep_pool_bwd(const ListPoolGradArgs* args, void* stream);

/* list_gemm_tn -- test/diagnostic entry for the transposed-operand MFMA kernel of the weight
 * gradients: out[M][N] = sum_p A[p][m] * B[p][n], A [P][M] and B [P][N] given as 16-bit planes like
 * list_gemm_nt.  M % 256 == 0, N % 8 == 0, P % 256 == 0. */
int list_gemm_tn(const void* a_hi, const void* a_lo, const void* b_hi, const void* b_lo, float* out,
                 void* slab, size_t slab_bytes, int32_t M, int32_t N, int32_t P, int32_t precision,
                 void* stream);

const char* list_last_error(void);
int list_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* LIST_HIP_H */
