"""Multi-threaded CPU restatement of the hot path as a torch-op sequence (TEST INFRASTRUCTURE).

This is the "reference CPU PyTorch path" timed by bench.py's cpu_baseline leg
(kind "port"): the op sequence that the reference executes at
network/modules.py:24-54 (interpolate -> matmul/div/clamp -> grid_sample x5 -> cat)
and network/modules.py:255-282 (stencil -> grid_sample 3-D x6 -> reshape/cat ->
Conv1d x4 + ReLU), fp32, no_grad.  It is pinned against the same golden vectors
as oracle/list_oracle.py (tests/test_oracle_golden.py).

Never imported by the product package.
"""
import torch
import torch.nn.functional as F

_STENCIL = torch.tensor([[0, 0, 0], [-1, 0, 0], [1, 0, 0], [0, -1, 0], [0, 1, 0],
                         [0, 0, -1], [0, 0, 1]], dtype=torch.float32) * 0.0722


def pooled_image_features(img_maps, pts, trans_mat, map_size=137, clamp_hi=136.0):
    """modules.py:24-54 -> [B,1024,N].  clamp_hi: the reference hard-codes 136 (modules.py:43); BASELINE config 5
    (map 274^2) clamps at map_size-1 instead."""
    ones = pts.new_ones(pts.shape[0], pts.shape[1], 1)
    cam = torch.matmul(torch.cat((pts, ones), dim=-1), trans_mat)
    uv = (cam[..., :2] / (cam[..., 2:] + 1e-8)).clamp(0.0, clamp_hi)
    half = (map_size - 1) / 2.0
    grid = ((uv - half) / half).unsqueeze(1)
    sampled = []
    for m in img_maps:
        up = F.interpolate(m, size=map_size, mode="bilinear", align_corners=True)
        sampled.append(F.grid_sample(up, grid, align_corners=True))
    return torch.cat(sampled, dim=1).squeeze(2)


def stencil_voxel_features(pts, vox_maps):
    """modules.py:256-273 -> [B,2583,N] with index c*7+j."""
    B, N, _ = pts.shape
    grid = (pts[:, None, None, :, :] + _STENCIL.to(pts)[None, None, :, None, :])   # [B,1,7,N,3]
    per_level = [F.grid_sample(f, grid, padding_mode="border", align_corners=True)
                 for f in vox_maps]
    stacked = torch.cat(per_level, dim=1)               # [B,369,1,7,N]
    return stacked.reshape(B, stacked.shape[1] * 7, N)


def implicit_mlp(features, weights):
    """modules.py:276-281."""
    h = features
    for name in ("fc_0", "fc_1", "fc_2"):
        h = F.relu(F.conv1d(h, weights[name + ".weight"], weights[name + ".bias"]))
    return F.conv1d(h, weights["fc_out.weight"], weights["fc_out.bias"]).squeeze(1)


@torch.no_grad()
def list_query(query, img_maps, vox_maps, trans_mat, weights, pre_permuted=False, map_size=137, clamp_hi=136.0):
    """models.py:91-97 -> sdf [B,N]."""
    pts = query if pre_permuted else query[:, :, [2, 1, 0]] * 2
    percep = pooled_image_features(img_maps, pts, trans_mat, map_size, clamp_hi)
    feats = torch.cat((stencil_voxel_features(pts, vox_maps), percep, pts.transpose(1, 2)), dim=1)
    return implicit_mlp(feats, weights)


def list_query_grads(query, img_maps, vox_maps, trans_mat, weights, grad_sdf, pre_permuted=False,
                     map_size=137, clamp_hi=136.0):
    """Backward of the path by autograd over the same op sequence (the reference trains through
    exactly these ops, train.py:82-85): gradients of sum(sdf * grad_sdf) w.r.t. the 2-D maps, the 3-D
    maps, trans_mat and the MLP parameters.  Returns (sdf, dict)."""
    with torch.enable_grad():
        leaf = lambda x: x.detach().clone().requires_grad_(True)
        img_l = [leaf(m) for m in img_maps]
        vox_l = [leaf(m) for m in vox_maps]
        T = leaf(trans_mat)
        W = {k: leaf(v) for k, v in weights.items()}
        pts = query if pre_permuted else query[:, :, [2, 1, 0]] * 2
        percep = pooled_image_features(img_l, pts, T, map_size, clamp_hi)
        feats = torch.cat((stencil_voxel_features(pts, vox_l), percep, pts.transpose(1, 2)), dim=1)
        sdf = implicit_mlp(feats, W)
        (sdf * grad_sdf).sum().backward()
    grads = {"d_trans_mat": T.grad}
    grads.update({f"d_img{i}": m.grad for i, m in enumerate(img_l)})
    grads.update({f"d_vox{i}": m.grad for i, m in enumerate(vox_l)})
    grads.update({"d_" + k: v.grad for k, v in W.items()})
    return sdf.detach(), grads


def to_torch(case):
    """numpy case dict (oracle/cases.py) -> torch tensors."""
    t = lambda a: torch.from_numpy(a.copy())
    return (t(case["query"]), [t(m) for m in case["img_maps"]], [t(m) for m in case["vox_maps"]],
            t(case["trans_mat"]), {k: t(v) for k, v in case["weights"].items()})
