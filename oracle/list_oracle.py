"""CPU restatement of LIST's SDF query hot path (TEST INFRASTRUCTURE -- the checker).

Plain numpy, float32 arithmetic, explicit formulas: no torch op is called, so
this file states the algorithm itself (what PyTorch's interpolate / grid_sample /
Conv1d compute at the reference's call sites).  Every function cites the
reference lines it follows (paths relative to /root/reference).

Parity status: PINNED -- tests/test_oracle_golden.py checks every function here
against tests/golden/*.npz, which oracle/gen_golden.py produced by running the
reference's own modules (network.modules.PerceptualPooling / VoxelDecoder2,
network.losses.SDFLoss, utils.create_grid_points_from_bounds) in the authoring
container (torch 2.10.0 CPU).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.
"""
import numpy as np

F32 = np.float32
MAP_SIZE = 137            # PerceptualPooling default, network/modules.py:16
DISPLACEMENT = 0.0722     # network/modules.py:205
VOX_CHANNELS = (1, 16, 32, 64, 128, 128)
IMG_CHANNELS = (64, 64, 128, 256, 512)


# --------------------------------------------------------------------------- a1
def permute_scale_query(query):
    """network/models.py:91-92: query[:, :, [2,1,0]] * 2."""
    q = np.asarray(query, dtype=F32)
    return (q[:, :, [2, 1, 0]] * F32(2.0)).astype(F32)


# --------------------------------------------------------------------------- a2
def resize_bilinear_align_corners(x, out_size):
    """F.interpolate(x, size, mode='bilinear', align_corners=True), network/modules.py:26-35.

    src = dst * (in-1)/(out-1); i0 = floor(src); i1 = min(i0+1, in-1); lerp in x then y."""
    x = np.asarray(x, dtype=F32)
    B, C, H, W = x.shape

    def axis(n_in):
        scale = F32(n_in - 1) / F32(out_size - 1) if out_size > 1 else F32(0)
        src = (np.arange(out_size, dtype=F32) * scale).astype(F32)
        i0 = np.minimum(src.astype(np.int64), n_in - 1)
        i1 = np.minimum(i0 + 1, n_in - 1)
        w1 = (src - i0.astype(F32)).astype(F32)
        w0 = (F32(1.0) - w1).astype(F32)
        return i0, i1, w0, w1

    y0, y1, wy0, wy1 = axis(H)
    x0, x1, wx0, wx1 = axis(W)
    top = x[:, :, y0][:, :, :, x0] * wx0 + x[:, :, y0][:, :, :, x1] * wx1
    bot = x[:, :, y1][:, :, :, x0] * wx0 + x[:, :, y1][:, :, :, x1] * wx1
    return (top * wy0[:, None] + bot * wy1[:, None]).astype(F32)


def project_points(pc, trans_mat, map_size=MAP_SIZE, clamp_hi=136.0):
    """network/modules.py:37-47: homogeneous matmul, perspective divide (+1e-8),
    clamp to [0,136] (hard-coded in the reference; `clamp_hi` generalises it for larger maps,
    BASELINE config 5), normalise by (map_size-1)/2."""
    pc = np.asarray(pc, dtype=F32)
    T = np.asarray(trans_mat, dtype=F32)
    # torch.matmul on CPU evaluates the K=4 dot product as an fma chain in k order
    # (checked bit-for-bit against torch 2.10 sgemm): x0*T0, then fma(x1,T1,.), fma(x2,T2,.),
    # fma(1,T3,.).  A float32 product is exact in float64, so one f64 add + rounding is an fma.
    def fma(a, b, acc):
        return (a.astype(np.float64) * b.astype(np.float64) + acc.astype(np.float64)).astype(F32)

    xyz = (pc[:, :, 0:1] * T[:, None, 0, :]).astype(F32)
    xyz = fma(pc[:, :, 1:2], T[:, None, 1, :], xyz)
    xyz = fma(pc[:, :, 2:3], T[:, None, 2, :], xyz)
    xyz = fma(np.ones_like(pc[:, :, 0:1]), T[:, None, 3, :], xyz)
    with np.errstate(divide="ignore", invalid="ignore"):
        xy = (xyz[:, :, :2] / (xyz[:, :, 2:3] + F32(1e-8))).astype(F32)
    # torch.clamp propagates NaN, np.clip does too
    xy = np.clip(xy, F32(0.0), F32(clamp_hi)).astype(F32)
    half = F32((map_size - 1) / 2.0)
    grid = ((xy - half) / half).astype(F32)
    return xy, grid


def grid_sample_2d(f, grid):
    """F.grid_sample(f, grid[B,1,N,2]) bilinear / zeros / align_corners=True,
    network/modules.py:48-52.  grid[...,0] -> width, grid[...,1] -> height.
    f: [B,C,H,W]; grid: [B,N,2]; returns [B,C,N]."""
    f = np.asarray(f, dtype=F32)
    g = np.asarray(grid, dtype=F32)
    B, C, H, W = f.shape
    ix = ((g[..., 0] + F32(1)) * F32((W - 1) / 2.0)).astype(F32)
    iy = ((g[..., 1] + F32(1)) * F32((H - 1) / 2.0)).astype(F32)
    x0f, y0f = np.floor(ix), np.floor(iy)
    x0, y0 = x0f.astype(np.int64), y0f.astype(np.int64)
    x1, y1 = x0 + 1, y0 + 1
    wx1 = (ix - x0f).astype(F32)
    wx0 = ((x0f + F32(1)) - ix).astype(F32)
    wy1 = (iy - y0f).astype(F32)
    wy0 = ((y0f + F32(1)) - iy).astype(F32)
    out = np.zeros((B, C, g.shape[1]), dtype=F32)
    bi = np.arange(B)[:, None]

    def tap(yy, xx, w):
        ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
        v = f[bi, :, np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)]      # [B,N,C]
        v = np.where(ok[..., None], v, F32(0))
        return np.transpose(v * w[..., None], (0, 2, 1)).astype(F32)

    out += tap(y0, x0, (wx0 * wy0).astype(F32))
    out += tap(y0, x1, (wx1 * wy0).astype(F32))
    out += tap(y1, x0, (wx0 * wy1).astype(F32))
    out += tap(y1, x1, (wx1 * wy1).astype(F32))
    return out


def perceptual_pooling(img_featuremaps, pc, trans_mat, map_size=MAP_SIZE, clamp_hi=136.0):
    """PerceptualPooling.forward, network/modules.py:24-54 -> [B,1024,1,N]."""
    _, grid = project_points(pc, trans_mat, map_size, clamp_hi)
    outs = [grid_sample_2d(resize_bilinear_align_corners(m, map_size), grid)
            for m in img_featuremaps]
    return np.concatenate(outs, axis=1)[:, :, None, :]


# --------------------------------------------------------------------------- a3/a4
def stencil():
    """network/modules.py:205-214: centre, then -d,+d per axis x,y,z."""
    d = [[0.0, 0.0, 0.0]]
    for ax in range(3):
        for s in (-1, 1):
            row = [0.0, 0.0, 0.0]
            row[ax] = s * DISPLACEMENT
            d.append(row)
    return np.asarray(d, dtype=F32)


def grid_sample_3d_border(f, p):
    """F.grid_sample(f, p, padding_mode='border', align_corners=True) for 5-D input,
    network/modules.py:263-265.  p[...,0]->W, p[...,1]->H, p[...,2]->D.
    f: [B,C,D,H,W]; p: [B,M,3]; returns [B,C,M].
    ix = ((x+1)/2)*(W-1), clipped to [0,W-1] BEFORE floor; taps whose index == size are skipped."""
    f = np.asarray(f, dtype=F32)
    p = np.asarray(p, dtype=F32)
    B, C, D, H, W = f.shape

    def unnorm(c, size):
        v = (((c + F32(1)) / F32(2)) * F32(size - 1)).astype(F32)
        # ATen's CPU clip_coordinates is std::min(size-1, std::max(v, 0)): every comparison with a NaN is
        # false, so a NaN coordinate comes out as size-1 (pinned by tests/golden/hotpath_edge_nan.npz)
        clipped = np.minimum(F32(size - 1), np.maximum(v, F32(0))).astype(F32)
        return np.where(np.isnan(v), F32(size - 1), clipped).astype(F32)

    ix, iy, iz = unnorm(p[..., 0], W), unnorm(p[..., 1], H), unnorm(p[..., 2], D)
    x0f, y0f, z0f = np.floor(ix), np.floor(iy), np.floor(iz)
    x0, y0, z0 = x0f.astype(np.int64), y0f.astype(np.int64), z0f.astype(np.int64)
    fx1, fy1, fz1 = (ix - x0f).astype(F32), (iy - y0f).astype(F32), (iz - z0f).astype(F32)
    fx0 = ((x0f + F32(1)) - ix).astype(F32)
    fy0 = ((y0f + F32(1)) - iy).astype(F32)
    fz0 = ((z0f + F32(1)) - iz).astype(F32)
    out = np.zeros((B, C, p.shape[1]), dtype=F32)
    bi = np.arange(B)[:, None]
    # accumulation order of the ATen CPU kernel: tnw tne tsw tse bnw bne bsw bse
    for dz, wz in ((0, fz0), (1, fz1)):
        for dy, wy in ((0, fy0), (1, fy1)):
            for dx, wx in ((0, fx0), (1, fx1)):
                xx, yy, zz = x0 + dx, y0 + dy, z0 + dz
                ok = (xx < W) & (yy < H) & (zz < D)
                v = f[bi, :, np.minimum(zz, D - 1), np.minimum(yy, H - 1), np.minimum(xx, W - 1)]
                v = np.where(ok[..., None], v, F32(0))
                w = ((wx * wy).astype(F32) * wz).astype(F32)
                out += np.transpose(v * w[..., None], (0, 2, 1)).astype(F32)
    return out


def vox_features(p, feat):
    """network/modules.py:256-273: 7-point stencil, 6 trilinear samples, feature index
    k = c*7 + j (c over the concatenated 369 channels, j the displacement) -> [B,2583,N]."""
    p = np.asarray(p, dtype=F32)
    B, N, _ = p.shape
    disp = stencil()
    pts = np.concatenate([(p + disp[j][None, None, :]).astype(F32) for j in range(7)], axis=1)
    per_level = []
    for f in feat:
        s = grid_sample_3d_border(f, pts)                 # [B,C,7N]
        per_level.append(s.reshape(B, f.shape[1], 7, N))
    allf = np.concatenate(per_level, axis=1)              # [B,369,7,N]
    return allf.reshape(B, allf.shape[1] * 7, N)


def mlp(features, weights, return_hidden=False):
    """network/modules.py:276-281: Conv1d(k=1) x4 with ReLU x3; features [B,K,N]."""
    h = np.asarray(features, dtype=F32)
    hidden = []
    for name in ("fc_0", "fc_1", "fc_2", "fc_out"):
        w = np.asarray(weights[name + ".weight"], dtype=F32)[:, :, 0]
        b = np.asarray(weights[name + ".bias"], dtype=F32)
        h = (np.einsum("ok,bkn->bon", w, h, optimize=True).astype(F32) + b[None, :, None]).astype(F32)
        if name != "fc_out":
            h = np.maximum(h, F32(0))
            hidden.append(h)
    out = h[:, 0, :]
    return (out, hidden) if return_hidden else out


def concat_features(p, feat, percep_feat):
    """network/modules.py:257,275: [2583 vox | 1024 perceptual | 3 coords] -> [B,3610,N]."""
    p = np.asarray(p, dtype=F32)
    return np.concatenate([vox_features(p, feat), np.asarray(percep_feat, dtype=F32),
                           np.transpose(p, (0, 2, 1))], axis=1)


def voxel_decoder2(p, feat, percep_feat, weights):
    """VoxelDecoder2.forward, network/modules.py:255-282 -> [B,N]."""
    return mlp(concat_features(p, feat, percep_feat), weights)


def list_query(query, img_featuremaps, vox_feat, trans_mat, weights, pre_permuted=False,
               map_size=MAP_SIZE, clamp_hi=136.0):
    """The per-point part of LIST.forward, network/models.py:91-97 -> sdf [B,N]."""
    q = np.asarray(query, dtype=F32) if pre_permuted else permute_scale_query(query)
    B, N, _ = q.shape
    percep = perceptual_pooling(img_featuremaps, q, trans_mat, map_size, clamp_hi).reshape(B, -1, N)
    return voxel_decoder2(q, vox_feat, percep, weights)


# --------------------------------------------------------------------------- a6 / losses
def create_grid_points_from_bounds(minimum, maximum, res):
    """utils.py:84-95: linspace, meshgrid 'ij' (X slowest), float64 [res^3,3]."""
    x = np.linspace(minimum, maximum, res)
    X, Y, Z = np.meshgrid(x, x, x, indexing="ij")
    return np.column_stack((X.reshape(-1), Y.reshape(-1), Z.reshape(-1)))


def sdf_loss(outputs, targets, sdf_scale=1.0):
    """network/losses.py:15-38."""
    o = np.asarray(outputs, dtype=F32)
    t = np.asarray(targets, dtype=F32)
    loss = np.mean(np.sum((t * F32(sdf_scale) - o) ** 2, axis=-1, dtype=F32), dtype=F32)
    real = np.mean((t - o / F32(sdf_scale)) ** 2, dtype=F32) * F32(10000)
    acc = np.mean(((t > 0.5) == (o > 0.5)).astype(F32), dtype=F32)
    return {"sdf_loss": loss, "ignore_sdf_loss_realvalue": real, "ignore_sdf_accuracy": acc}
