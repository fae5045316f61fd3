// Point-parallel feature gather (HBM/L2-bound; no MFMA):
//   * k_gather_vox<C>  : 7-point stencil x trilinear (border, align_corners) sample of one
//                        channels-last voxel level  (network/modules.py:256-265)
//   * k_gather_img     : projection by trans_mat, perspective divide, clamp, bilinear sample of
//                        the channels-last 137^2 map  (network/modules.py:37-52)
//   * k_gather_tail    : scalar (C==1) voxel levels, xyz coordinates, zero padding
// All write the bf16 hi/lo feature matrix X[row][Kp] in gather order (list_common.h).
// Lanes run over channel quads (16-B loads, coalesced along C); a workgroup owns 64 points.
#include "list_common.h"

namespace list {

// ---- shared per-point helpers -----------------------------------------------------------------
struct Pt { float x, y, z; int b; bool valid; };

__device__ __forceinline__ Pt load_point(const GatherParams& g, int row) {
  Pt p;
  p.valid = row < g.n_valid;
  const int64_t gp = g.p_begin + (p.valid ? (g.order ? g.order[row] : row) : 0);
  p.b = (int)(gp / g.N);
  const int n = (int)(gp - (int64_t)p.b * g.N);
  const float* q = g.query + (int64_t)p.b * g.q_sb + (int64_t)n * g.q_sn;
  p.x = q[(int64_t)g.perm0 * g.q_sc] * g.scale;     // models.py:91-92: query[:, :, [2,1,0]] * 2
  p.y = q[(int64_t)g.perm1 * g.q_sc] * g.scale;
  p.z = q[(int64_t)g.perm2 * g.q_sc] * g.scale;
  return p;
}

// grid_sampler_unnormalize (align_corners) + clip_coordinates + floor, as ATen computes them:
//   v = ((c + 1) / 2) * (size - 1); v = min(size-1, max(v, 0)); i0 = floor(v)
//   w1 = v - i0 ; w0 = (i0 + 1) - v ; the +1 tap is skipped when i0 + 1 == size (its weight is 0)
struct Axis { int i0; int has1; float w0, w1; };
__device__ __forceinline__ Axis axis_setup(float c, int size) {
  float v = ((c + 1.f) * 0.5f) * (float)(size - 1);
  v = fminf((float)(size - 1), fmaxf(v, 0.f));
  const float f = floorf(v);
  Axis a;
  a.i0 = (int)f;
  a.has1 = (a.i0 + 1 < size) ? 1 : 0;
  a.w1 = v - f;
  a.w0 = (f + 1.f) - v;
  return a;
}

__device__ __forceinline__ float4 fma4(const float4& v, float w, const float4& a) {
  return make_float4(fmaf(v.x, w, a.x), fmaf(v.y, w, a.y), fmaf(v.z, w, a.z), fmaf(v.w, w, a.w));
}

// One trilinear sample of 4 channels.  base points at (image, channel quad).
// Accumulation order of the ATen CPU kernel: tnw tne tsw tse bnw bne bsw bse.
__device__ __forceinline__ float4 trilinear4(const float* __restrict__ base, int C, int H, int W,
                                             const Axis& ax, const Axis& ay, const Axis& az) {
  const int o000 = ((az.i0 * H + ay.i0) * W + ax.i0) * C;
  const int sx = ax.has1 ? C : 0;
  const int sy = ay.has1 ? W * C : 0;
  const int sz = az.has1 ? H * W * C : 0;
  const float4 v000 = *(const float4*)(base + o000);
  const float4 v001 = *(const float4*)(base + o000 + sx);
  const float4 v010 = *(const float4*)(base + o000 + sy);
  const float4 v011 = *(const float4*)(base + o000 + sy + sx);
  const float4 v100 = *(const float4*)(base + o000 + sz);
  const float4 v101 = *(const float4*)(base + o000 + sz + sx);
  const float4 v110 = *(const float4*)(base + o000 + sz + sy);
  const float4 v111 = *(const float4*)(base + o000 + sz + sy + sx);
  const float tnw = ax.w0 * ay.w0 * az.w0, tne = ax.w1 * ay.w0 * az.w0;
  const float tsw = ax.w0 * ay.w1 * az.w0, tse = ax.w1 * ay.w1 * az.w0;
  const float bnw = ax.w0 * ay.w0 * az.w1, bne = ax.w1 * ay.w0 * az.w1;
  const float bsw = ax.w0 * ay.w1 * az.w1, bse = ax.w1 * ay.w1 * az.w1;
  float4 acc = make_float4(v000.x * tnw, v000.y * tnw, v000.z * tnw, v000.w * tnw);
  acc = fma4(v001, tne, acc);
  acc = fma4(v010, tsw, acc);
  acc = fma4(v011, tse, acc);
  acc = fma4(v100, bnw, acc);
  acc = fma4(v101, bne, acc);
  acc = fma4(v110, bsw, acc);
  acc = fma4(v111, bse, acc);
  return acc;
}

// stencil point j of network/modules.py:205-214: centre, then (-d,+d) along x, y, z
template <int J>
__device__ __forceinline__ void stencil_point(const Pt& p, float& x, float& y, float& z) {
  x = p.x + (J == 1 ? -kDisp : J == 2 ? kDisp : 0.f);
  y = p.y + (J == 3 ? -kDisp : J == 4 ? kDisp : 0.f);
  z = p.z + (J == 5 ? -kDisp : J == 6 ? kDisp : 0.f);
}

template <int C, int J, int FMT>
__device__ __forceinline__ void gather_one(const GatherParams& g, const ListVoxLevel& lv,
                                           const float* __restrict__ base, const Pt& p,
                                           int64_t out_off) {
  float x, y, z;
  stencil_point<J>(p, x, y, z);
  const Axis ax = axis_setup(x, lv.W), ay = axis_setup(y, lv.H), az = axis_setup(z, lv.D);
  float4 v = trilinear4(base, C, lv.H, lv.W, ax, ay, az);
  if (!p.valid) v = make_float4(0.f, 0.f, 0.f, 0.f);
  store_feat4<FMT>(g.x_hi, g.x_lo, out_off + J * C, v);
}

// grid = rows/64, block = 256.  LP = C/4 lanes share a point; a wave covers 64/LP points.
template <int C, int FMT>
__global__ __launch_bounds__(256) void k_gather_vox(GatherParams g, ListVoxLevel lv, int col_off) {
  constexpr int LP = C / 4;              // lanes per point
  constexpr int PW = 64 / LP;            // points per wave per iteration
  constexpr int ITERS = (4 * PW >= kGatherRows) ? 1 : kGatherRows / (4 * PW);
  constexpr int RB = 4 * PW * ITERS;     // rows per workgroup (64, 128 or 256: divides g.rows)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int quad = lane % LP, psub = lane / LP;
  const int blk = xcd_contiguous_block(blockIdx.x, gridDim.x);
#pragma unroll 1
  for (int it = 0; it < ITERS; ++it) {
    const int row = blk * RB + it * (4 * PW) + wave * PW + psub;
    const Pt p = load_point(g, row);
    const float* base = lv.data + (int64_t)p.b * lv.image_stride + quad * 4;
    const int64_t out_off = (int64_t)row * g.Kp + col_off + quad * 4;
    gather_one<C, 0, FMT>(g, lv, base, p, out_off);
    gather_one<C, 1, FMT>(g, lv, base, p, out_off);
    gather_one<C, 2, FMT>(g, lv, base, p, out_off);
    gather_one<C, 3, FMT>(g, lv, base, p, out_off);
    gather_one<C, 4, FMT>(g, lv, base, p, out_off);
    gather_one<C, 5, FMT>(g, lv, base, p, out_off);
    gather_one<C, 6, FMT>(g, lv, base, p, out_off);
  }
}

// ---- 2-D perceptual pooling ---------------------------------------------------------------------
// network/modules.py:37-47 per point.  torch.matmul evaluates the K=4 dot product as an fma chain
// in k order (oracle/list_oracle.py project_points, checked bit-for-bit).
struct Proj { int o00, o01, o10, o11; float w00, w01, w10, w11; };

__device__ __forceinline__ float clamp_keep_nan(float v, float hi) {
  return (v != v) ? v : fminf(fmaxf(v, 0.f), hi);
}

__device__ __forceinline__ Proj project(const float* __restrict__ T, float px, float py, float pz,
                                        int ms, int Ct, float clamp_hi) {
  float X = px * T[0], Y = px * T[1], Z = px * T[2];
  X = fmaf(py, T[3], X); Y = fmaf(py, T[4], Y); Z = fmaf(py, T[5], Z);
  X = fmaf(pz, T[6], X); Y = fmaf(pz, T[7], Y); Z = fmaf(pz, T[8], Z);
  X = X + T[9]; Y = Y + T[10]; Z = Z + T[11];
  const float den = Z + 1e-8f;
  float u = clamp_keep_nan(__fdiv_rn(X, den), clamp_hi);
  float v = clamp_keep_nan(__fdiv_rn(Y, den), clamp_hi);
  const float half = (float)(ms - 1) * 0.5f;
  const float gx = __fdiv_rn(u - half, half), gy = __fdiv_rn(v - half, half);
  const float ix = (gx + 1.f) * half, iy = (gy + 1.f) * half;     // grid_sample unnormalize
  const float fx = floorf(ix), fy = floorf(iy);
  const float wx1 = ix - fx, wx0 = (fx + 1.f) - ix;
  const float wy1 = iy - fy, wy0 = (fy + 1.f) - iy;
  // zeros padding: a tap at index ms only occurs with weight 0 (ix == ms-1); clamp for safety
  const int x0 = min(max((int)fx, 0), ms - 1), y0 = min(max((int)fy, 0), ms - 1);
  const int x1 = min(x0 + 1, ms - 1), y1 = min(y0 + 1, ms - 1);
  Proj r;
  r.o00 = (y0 * ms + x0) * Ct; r.o01 = (y0 * ms + x1) * Ct;
  r.o10 = (y1 * ms + x0) * Ct; r.o11 = (y1 * ms + x1) * Ct;
  r.w00 = wx0 * wy0; r.w01 = wx1 * wy0; r.w10 = wx0 * wy1; r.w11 = wx1 * wy1;
  return r;
}

__device__ __forceinline__ float4 bilinear4(const float* __restrict__ img, const Proj& pr) {
  const float4 v00 = *(const float4*)(img + pr.o00);
  const float4 v01 = *(const float4*)(img + pr.o01);
  const float4 v10 = *(const float4*)(img + pr.o10);
  const float4 v11 = *(const float4*)(img + pr.o11);
  float4 acc = make_float4(v00.x * pr.w00, v00.y * pr.w00, v00.z * pr.w00, v00.w * pr.w00);
  acc = fma4(v01, pr.w01, acc);
  acc = fma4(v10, pr.w10, acc);
  acc = fma4(v11, pr.w11, acc);
  return acc;
}

// grid = rows/64, block = 256: the block walks its 64 points, lanes over channel quads.
template <int FMT>
__global__ __launch_bounds__(256) void k_gather_img(GatherParams g, const float* __restrict__ img_map,
                                                    const float* __restrict__ trans_mat, int ms,
                                                    int Ct, float clamp_hi, int col_off) {
  const int nq = Ct / 4;
  const int blk = xcd_contiguous_block(blockIdx.x, gridDim.x);
#pragma unroll 2
  for (int i = 0; i < kGatherRows; ++i) {
    const int row = blk * kGatherRows + i;
    const Pt p = load_point(g, row);
    const Proj pr = project(trans_mat + p.b * 12, p.x, p.y, p.z, ms, Ct, clamp_hi);
    const float* img = img_map + (int64_t)p.b * ms * ms * Ct;
    for (int q = threadIdx.x; q < nq; q += 256) {
      float4 v = bilinear4(img + q * 4, pr);
      if (!p.valid) v = make_float4(0.f, 0.f, 0.f, 0.f);
      store_feat4<FMT>(g.x_hi, g.x_lo, (int64_t)row * g.Kp + col_off + q * 4, v);
    }
  }
}

// Pre-pooled perceptual features [B,img_C,N] (VoxelDecoder2.forward's third argument) -> X.
// lane = row so reads are coalesced along N.
template <int FMT>
__global__ __launch_bounds__(256) void k_copy_percep(GatherParams g, const float* __restrict__ pf,
                                                     int64_t sb, int64_t sc, int64_t sn, int Ct,
                                                     int col_off) {
  const int row = blockIdx.x * 256 + threadIdx.x;
  if (row >= g.rows) return;
  const bool valid = row < g.n_valid;
  const int64_t gp = g.p_begin + (valid ? (g.order ? g.order[row] : row) : 0);
  const int b = (int)(gp / g.N);
  const int n = (int)(gp - (int64_t)b * g.N);
  const float* src = pf + (int64_t)b * sb + (int64_t)n * sn;
  for (int c = 0; c < Ct; c += 4) {
    float4 v = make_float4(src[(int64_t)c * sc], src[(int64_t)(c + 1) * sc],
                           src[(int64_t)(c + 2) * sc], src[(int64_t)(c + 3) * sc]);
    if (!valid) v = make_float4(0.f, 0.f, 0.f, 0.f);
    store_feat4<FMT>(g.x_hi, g.x_lo, (int64_t)row * g.Kp + col_off + c, v);
  }
}

// ---- tail: scalar voxel levels, xyz, zero padding -----------------------------------------------
struct TailLevels { ListVoxLevel lv[LIST_N_VOX_LEVELS]; int off[LIST_N_VOX_LEVELS]; int n; };

__device__ __forceinline__ float trilinear1(const float* __restrict__ base, int H, int W,
                                            const Axis& ax, const Axis& ay, const Axis& az) {
  const int o = (az.i0 * H + ay.i0) * W + ax.i0;
  const int sx = ax.has1 ? 1 : 0, sy = ay.has1 ? W : 0, sz = az.has1 ? H * W : 0;
  float acc = base[o] * (ax.w0 * ay.w0 * az.w0);
  acc = fmaf(base[o + sx], ax.w1 * ay.w0 * az.w0, acc);
  acc = fmaf(base[o + sy], ax.w0 * ay.w1 * az.w0, acc);
  acc = fmaf(base[o + sy + sx], ax.w1 * ay.w1 * az.w0, acc);
  acc = fmaf(base[o + sz], ax.w0 * ay.w0 * az.w1, acc);
  acc = fmaf(base[o + sz + sx], ax.w1 * ay.w0 * az.w1, acc);
  acc = fmaf(base[o + sz + sy], ax.w0 * ay.w1 * az.w1, acc);
  acc = fmaf(base[o + sz + sy + sx], ax.w1 * ay.w1 * az.w1, acc);
  return acc;
}

template <int FMT>
__device__ __forceinline__ void put(const GatherParams& g, int64_t o, float v) {
  store_feat1<FMT>(g.x_hi, g.x_lo, o, v);
}

template <int FMT>
__global__ __launch_bounds__(256) void k_gather_tail(GatherParams g, TailLevels tl, int xyz_off,
                                                     int F) {
  const int row = blockIdx.x * 256 + threadIdx.x;
  if (row >= g.rows) return;
  const Pt p = load_point(g, row);
  const int64_t ro = (int64_t)row * g.Kp;
  for (int l = 0; l < tl.n; ++l) {
    const ListVoxLevel& lv = tl.lv[l];
    const float* base = lv.data + (int64_t)p.b * lv.image_stride;
#pragma unroll
    for (int j = 0; j < LIST_N_STENCIL; ++j) {
      const float x = p.x + (j == 1 ? -kDisp : j == 2 ? kDisp : 0.f);
      const float y = p.y + (j == 3 ? -kDisp : j == 4 ? kDisp : 0.f);
      const float z = p.z + (j == 5 ? -kDisp : j == 6 ? kDisp : 0.f);
      const Axis ax = axis_setup(x, lv.W), ay = axis_setup(y, lv.H), az = axis_setup(z, lv.D);
      const float v = trilinear1(base, lv.H, lv.W, ax, ay, az);
      put<FMT>(g, ro + tl.off[l] + j, p.valid ? v : 0.f);
    }
  }
  put<FMT>(g, ro + xyz_off + 0, p.valid ? p.x : 0.f);     // p_features, modules.py:257
  put<FMT>(g, ro + xyz_off + 1, p.valid ? p.y : 0.f);
  put<FMT>(g, ro + xyz_off + 2, p.valid ? p.z : 0.f);
  for (int k = F; k < g.Kp; ++k) {
    g.x_hi[ro + k] = 0;
    if (FMT == FMT_BF16_SPLIT) g.x_lo[ro + k] = 0;
  }
}

// ---- Morton ordering of the query points -------------------------------------------------------------
// Random query points make every tap a cold line.  Rows are therefore processed in (image, Morton
// cell) order, a counting sort on key = (b % 64) * 4096 + morton(16^3 cell): consecutive rows (and so
// the workgroups resident at any moment, on every XCD) sample one small region of the maps, which
// turns most taps into L1/L2 hits.  Results are independent of the row order bit for bit (every row
// is computed on its own); the order inside a cell comes from atomics and is not deterministic.
__device__ __forceinline__ unsigned spread3(unsigned v) {      // 4 bits -> every third bit
  v = (v | (v << 4)) & 0x0C3u;
  v = (v | (v << 2)) & 0x249u;
  return v;
}
__device__ __forceinline__ int sort_key(const GatherParams& g, int i, int b_first) {
  const int64_t gp = g.p_begin + i;
  const int b = (int)(gp / g.N);
  const int n = (int)(gp - (int64_t)b * g.N);
  const float* q = g.query + (int64_t)b * g.q_sb + (int64_t)n * g.q_sn;
  unsigned c[3];
  const int perm[3] = {g.perm0, g.perm1, g.perm2};
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float v = (q[(int64_t)perm[a] * g.q_sc] * g.scale + 1.f) * (0.5f * kSortCellsPerAxis);
    c[a] = (unsigned)fminf(fmaxf(v, 0.f), (float)(kSortCellsPerAxis - 1));
  }
  const unsigned m = spread3(c[0]) | (spread3(c[1]) << 1) | (spread3(c[2]) << 2);
  return ((b - b_first) % kSortImages) * kSortCells + (int)m;
}

__global__ __launch_bounds__(256) void k_sort_hist(GatherParams g, int b_first, int* __restrict__ keys,
                                                   int* __restrict__ bins) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= g.n_valid) return;
  const int k = sort_key(g, i, b_first);
  keys[i] = k;
  atomicAdd(&bins[k], 1);
}

// exclusive scan of nbins (a multiple of 4096) counters by ONE workgroup of 1024 threads
__global__ __launch_bounds__(1024) void k_sort_scan(int* __restrict__ bins, int nbins) {
  __shared__ int part[1024];
  const int PER = nbins / 1024;
  int* mine = bins + threadIdx.x * PER;
  int sum = 0;
  for (int i = 0; i < PER; ++i) sum += mine[i];
  part[threadIdx.x] = sum;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const int v = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  int run = part[threadIdx.x] - sum;
  for (int i = 0; i < PER; ++i) { const int c = mine[i]; mine[i] = run; run += c; }
}

__global__ __launch_bounds__(256) void k_sort_scatter(int n_valid, const int* __restrict__ keys,
                                                      int* __restrict__ bins, int* __restrict__ order) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_valid) return;
  order[atomicAdd(&bins[keys[i]], 1)] = i;
}

hipError_t launch_sort_points(const GatherParams& g, int* order, int* keys, int* bins, hipStream_t s) {
  const int b_first = (int)(g.p_begin / g.N);
  const int64_t b_last = (g.p_begin + g.n_valid - 1) / g.N;
  const int nslots = (int)((b_last - b_first + 1 < kSortImages) ? (b_last - b_first + 1) : kSortImages);
  const int nbins = nslots * kSortCells;
  hipError_t e = hipMemsetAsync(bins, 0, (size_t)nbins * sizeof(int), s);
  if (e != hipSuccess) return e;
  const unsigned nb = (unsigned)((g.n_valid + 255) / 256);
  GatherParams raw = g;
  raw.order = nullptr;
  hipLaunchKernelGGL(k_sort_hist, dim3(nb), dim3(256), 0, s, raw, b_first, keys, bins);
  hipLaunchKernelGGL(k_sort_scan, dim3(1), dim3(1024), 0, s, bins, nbins);
  hipLaunchKernelGGL(k_sort_scatter, dim3(nb), dim3(256), 0, s, g.n_valid, keys, bins, order);
  return hipGetLastError();
}

// ---- launch ----------------------------------------------------------------------------------------
template <int C, int FMT>
static hipError_t launch_vox_level(const GatherParams& g, const ListVoxLevel& lv, int col_off,
                                   hipStream_t s) {
  constexpr int PW = 64 / (C / 4);
  constexpr int RB = (4 * PW >= kGatherRows) ? 4 * PW : kGatherRows;
  hipLaunchKernelGGL((k_gather_vox<C, FMT>), dim3(g.rows / RB), dim3(256), 0, s, g, lv, col_off);
  return hipGetLastError();
}

template <int FMT>
static hipError_t launch_gather_fmt(const GatherParams& g, const FeatLayout& L, const ListQueryArgs& a,
                                    hipStream_t s) {
  hipError_t e = hipSuccess;
  TailLevels tl;
  tl.n = 0;
  for (int l = 0; l < LIST_N_VOX_LEVELS; ++l) {
    const ListVoxLevel& lv = a.vox[l];
    if (lv.C == 1) { tl.lv[tl.n] = lv; tl.off[tl.n] = L.vox_off[l]; ++tl.n; continue; }
    switch (lv.C) {
      case 4: e = launch_vox_level<4, FMT>(g, lv, L.vox_off[l], s); break;
      case 8: e = launch_vox_level<8, FMT>(g, lv, L.vox_off[l], s); break;
      case 16: e = launch_vox_level<16, FMT>(g, lv, L.vox_off[l], s); break;
      case 32: e = launch_vox_level<32, FMT>(g, lv, L.vox_off[l], s); break;
      case 64: e = launch_vox_level<64, FMT>(g, lv, L.vox_off[l], s); break;
      case 128: e = launch_vox_level<128, FMT>(g, lv, L.vox_off[l], s); break;
      case 256: e = launch_vox_level<256, FMT>(g, lv, L.vox_off[l], s); break;
      default: return hipErrorInvalidValue;
    }
    if (e != hipSuccess) return e;
  }
  if (a.stage_events && a.stage_events[LIST_STAGE_VOX])
    (void)hipEventRecord((hipEvent_t)a.stage_events[LIST_STAGE_VOX], s);
  if (a.percep_feat) {
    hipLaunchKernelGGL(k_copy_percep<FMT>, dim3((g.rows + 255) / 256), dim3(256), 0, s, g,
                       a.percep_feat, a.pf_sb, a.pf_sc, a.pf_sn, L.img_C, L.img_off);
  } else {
    hipLaunchKernelGGL(k_gather_img<FMT>, dim3(g.rows / kGatherRows), dim3(256), 0, s, g, a.img_map,
                       a.trans_mat, a.map_size, L.img_C, a.clamp_hi, L.img_off);
  }
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_gather_tail<FMT>, dim3((g.rows + 255) / 256), dim3(256), 0, s, g, tl, L.xyz_off,
                     L.F);
  return hipGetLastError();
}

hipError_t launch_gather(const GatherParams& g, const FeatLayout& L, const ListQueryArgs& a,
                         hipStream_t s) {
  return g.fmt == FMT_FP16 ? launch_gather_fmt<FMT_FP16>(g, L, a, s)
                           : launch_gather_fmt<FMT_BF16_SPLIT>(g, L, a, s);
}

// ---- diagnostics: X (gather order, hi+lo) -> out[B][F][N] in the reference order ---------------------
__global__ __launch_bounds__(256) void k_features_out(GatherParams g, FeatLayout L,
                                                      float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)g.n_valid * L.Kp;
  if (i >= total) return;
  const int row = (int)(i / L.Kp), kp = (int)(i - (int64_t)row * L.Kp);
  const int kr = ref_index_of(L, kp);
  if (kr < 0) return;
  const int64_t gp = g.p_begin + (g.order ? g.order[row] : row);
  const int b = (int)(gp / g.N);
  const int n = (int)(gp - (int64_t)b * g.N);
  out[((int64_t)b * L.F + kr) * g.N + n] =
      g.fmt == FMT_FP16 ? h2f(g.x_hi[i]) : bf2f(g.x_hi[i]) + bf2f(g.x_lo[i]);
}

hipError_t launch_features_out(const GatherParams& g, const FeatLayout& L, float* out, int B,
                               hipStream_t s) {
  (void)B;
  const int64_t total = (int64_t)g.n_valid * L.Kp;
  hipLaunchKernelGGL(k_features_out, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, g, L,
                     out);
  return hipGetLastError();
}

// ---- PerceptualPooling.forward alone: out[B][Ct][N] ---------------------------------------------------
// lanes over points (coalesced writes along N), loop over channel quads.
__global__ __launch_bounds__(256) void k_percep_pool(ListPoolArgs a) {
  const int64_t gp = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gp >= (int64_t)a.B * a.N) return;
  const int b = (int)(gp / a.N);
  const int n = (int)(gp - (int64_t)b * a.N);
  const float* q = a.pc + (int64_t)b * a.p_sb + (int64_t)n * a.p_sn;
  const Proj pr = project(a.trans_mat + b * 12, q[0], q[a.p_sc], q[2 * a.p_sc], a.map_size,
                          a.img_C, a.clamp_hi);
  const float* img = a.img_map + (int64_t)b * a.map_size * a.map_size * a.img_C;
  float* o = a.out + (int64_t)b * a.img_C * a.N + n;
  for (int c = 0; c < a.img_C; c += 4) {
    const float4 v = bilinear4(img + c, pr);
    o[(int64_t)c * a.N] = v.x;
    o[(int64_t)(c + 1) * a.N] = v.y;
    o[(int64_t)(c + 2) * a.N] = v.z;
    o[(int64_t)(c + 3) * a.N] = v.w;
  }
}

hipError_t launch_percep_pool(const ListPoolArgs& a, hipStream_t s) {
  const int64_t P = (int64_t)a.B * a.N;
  hipLaunchKernelGGL(k_percep_pool, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, s, a);
  return hipGetLastError();
}

}  // namespace list
