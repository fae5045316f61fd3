"""Shared definitions of the parity cases (TEST INFRASTRUCTURE).

Each case is a dict of numpy inputs built only from oracle/synth.py, so the
golden generator (reference side), the oracle tests and the GPU parity tests all
regenerate bit-identical inputs without storing them.
"""
import functools

import numpy as np

from . import synth

F32 = np.float32


@functools.lru_cache(maxsize=None)
def build_case(name):
    """Cached: callers must treat the returned arrays as read-only."""
    if name == "tiny":          # B=2, odd N, 32^2 image, 16^3 voxels (coarsest map is 1^3)
        return _case(seed=101, batch=2, n=129, img_res=32, vox_res=16)
    if name == "small":         # B=3, 64^2 image, 32^3 voxels
        return _case(seed=202, batch=3, n=200, img_res=64, vox_res=32)
    if name == "real":          # the metric's map sizes, few points
        return _case(seed=333, batch=1, n=64, img_res=224, vox_res=128)
    if name == "edge":
        return _edge_case()
    if name == "edge_nan":
        return _edge_nan_case()
    # gradient cases: same shapes, seeds chosen so that no ReLU pre-activation of the fp32 forward is
    # closer to zero than 4e-6 (the mask is discontinuous: a sign flip from 1e-6-level arithmetic
    # differences would change the gradients by percents; searched with tools/find_margin_seeds.py)
    if name == "gtiny":
        return _case(seed=1116, batch=2, n=129, img_res=32, vox_res=16)
    if name == "gsmall":
        return _case(seed=2394, batch=3, n=67, img_res=64, vox_res=32)
    if name == "gedge":
        return _edge_case(seed=GEDGE_SEED)
    raise KeyError(name)


def _case(seed, batch, n, img_res, vox_res):
    return {
        "query": synth.make_query(seed, batch, n),
        "img_maps": synth.make_img_maps(seed, batch, img_res),
        "vox_maps": synth.make_vox_maps(seed, batch, vox_res),
        "trans_mat": synth.make_trans_mat(seed, batch),
        "weights": synth.make_mlp_weights(seed),
    }


def _edge_case(seed=404):
    """Points on the +-1 faces, projections clamped at 0 and at 136 (dropped tap),
    Z+1e-8 <= 0 (sign flip / inf -> clamp), displaced coordinates outside [-1,1]."""
    c = _case(seed=seed, batch=1, n=64, img_res=32, vox_res=16)
    q = c["query"].copy()
    corners = np.array([[s0, s1, s2] for s0 in (-0.5, 0.5) for s1 in (-0.5, 0.5)
                        for s2 in (-0.5, 0.5)], dtype=F32)
    q[0, :8] = corners                      # all eight cube corners (p = +-1 after *2)
    q[0, 8] = (0.5, 0.0, 0.0)
    q[0, 9] = (0.0, -0.5, 0.0)
    q[0, 10] = (0.0, 0.0, 0.5)
    q[0, 11] = (0.0, 0.0, 0.0)              # centre -> voxel coordinate exactly (W-1)/2
    q[0, 12] = (0.4639, -0.4639, 0.4639)    # displaced stencil points leave [-1,1]
    c["query"] = q
    # camera: u = 68 + 136*px', v = 68 - 136*py', Z = 1 + 2*pz'  (px' = permuted/scaled coords)
    # -> plenty of clamping at 0 and 136, and Z <= 0 for pz' <= -0.5
    T = np.array([[[136.0, 0.0, 0.0], [0.0, -136.0, 0.0], [0.0, 0.0, 2.0], [68.0, 68.0, 1.0]]],
                 dtype=F32)
    c["trans_mat"] = T
    return c


def _edge_nan_case(seed=606):
    """Non-finite inputs (the semantics ATen's CPU kernels give them are pinned by the golden):
    NaN query coordinates (3-D: clip_coordinates turns a NaN into size-1; 2-D: NaN propagates), a 0/0 projection,
    +-inf / NaN voxels and pixels -- at border taps the reference SKIPS (index == size), inside the
    shared-tap window of the coarse levels with weight 0 for some stencil samples, and at taps that are used.
    Special points sit at even indices (the fixture keeps every 2nd point's features)."""
    c = _case(seed=seed, batch=2, n=32, img_res=32, vox_res=32)
    q = c["query"].copy()
    nan = np.float32(np.nan)
    q[0, 0] = (nan, 0.1, 0.2)
    q[0, 2] = (0.1, nan, -0.2)
    q[0, 4] = (0.3, 0.2, nan)
    q[0, 6] = (nan, nan, nan)
    q[0, 8] = (0.5, 0.5, 0.5)              # p = +1 on every axis: the +1 taps are skipped
    q[0, 10] = (-0.5, 0.5, 0.0)
    q[0, 12] = (0.47, 0.0, 0.0)            # the +d stencil sample is clipped onto the border
    q[0, 14] = (-0.0667, -0.0667, -0.0667)  # next to the -inf voxel of the 16^3 level (window, weight 0 for some samples)
    q[0, 16] = (-0.48, -0.48, -0.48)       # next to the NaN voxel of the 8^3 level
    q[0, 18] = (0.0, 0.0, 0.0)
    q[1, 0] = (0.1, 0.2, -0.25)            # image 1: X = 0 and Z + 1e-8 = 0 -> u = 0/0
    q[1, 2] = (0.1, 0.2, 0.25)             # X / 0 -> +inf -> clamped
    c["query"] = q
    vox = [m.copy() for m in c["vox_maps"]]
    inf = np.float32(np.inf)
    vox[0][0, 0, 31, 31, 31] = inf         # C = 1 level, far corner (border tap of q[0,8])
    vox[1][0, 3, 31, 31, 31] = inf         # 32^3 x 16: generic kernel, border voxel
    vox[1][0, 5, 31, 31, 16] = -inf
    vox[2][0, 5, 8, 8, 8] = -inf           # 16^3 x 32: shared-tap kernel, interior voxel
    vox[3][0, 0, 0, 0, 0] = nan            # 8^3 x 64
    vox[4][0, 9, 3, 3, 3] = inf            # 4^3 x 128 far corner (every point of the last cell)
    vox[1][1, 2, 0, 0, 0] = inf            # image 1, near corner
    c["vox_maps"] = vox
    img = [m.copy() for m in c["img_maps"]]
    img[0][0, 3, 31, 31] = -inf            # 32^2 source: the resized pixels around (136, 136)
    img[1][1, 4, 0, 0] = inf               # 16^2 source, image 1: around (0, 0)
    c["img_maps"] = img
    T = np.array([[[136.0, 0.0, 0.0], [0.0, -136.0, 0.0], [0.0, 0.0, 2.0], [68.0, 68.0, 1.0]],
                  [[136.0, 0.0, 0.0], [0.0, -136.0, 0.0], [0.0, 0.0, 0.0], [68.0, 68.0, -1e-8]]], dtype=F32)
    c["trans_mat"] = T
    return c


GEDGE_SEED = 405
CASE_NAMES = ("tiny", "small", "real", "edge")
NONFINITE_CASE_NAMES = ("edge_nan",)       # fixture keeps every 2nd point's features (FEATURE_STRIDE)
FEATURE_STRIDE = {"edge_nan": 2}
GRAD_CASE_NAMES = ("gtiny", "gsmall", "gedge")
