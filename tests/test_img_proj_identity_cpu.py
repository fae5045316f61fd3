"""CPU: the identity list_prep_img_proj (ABI 8) rests on, checked with the ORACLE's own restatements of the reference ops
(oracle/list_oracle.py: resize_bilinear_align_corners = network/modules.py:26-35, grid_sample_2d = :48-52, mlp = :276-281):

    fc_0-columns(sample(resize(x_l)))  ==  sample(resize(fc_0-columns(x_l)))      for every encoder level l,

because F.interpolate and fc_0 are linear and the resize applies the same weights to every channel.  The right-hand side
is what the HIP path evaluates for the projected levels (one product per source pixel instead of one per query point);
the test also pins the rule that chooses them (hip.img_proj_kept_levels) on the reference's pyramids."""
import numpy as np

from oracle import cases, list_oracle as O


def test_projecting_a_level_before_the_resize_is_the_same_field():
    c = cases.build_case("small")
    q = O.permute_scale_query(c["query"])
    _, grid = O.project_points(q, c["trans_mat"])
    W0 = c["weights"]["fc_0.weight"].reshape(512, -1)
    F = W0.shape[1]
    img_C = sum(m.shape[1] for m in c["img_maps"])
    col = F - 3 - img_C                                   # the perceptual block sits before xyz in the reference order
    B, N = q.shape[:2]
    for x in c["img_maps"]:
        C = x.shape[1]
        w = W0[:, col:col + C].astype(np.float64)         # [512, C]
        # reference order of operations: resize, sample, multiply by the level's columns of fc_0
        feat = O.grid_sample_2d(O.resize_bilinear_align_corners(x, O.MAP_SIZE), grid)            # [B, C, N]
        ref = np.einsum("nc,bcp->bnp", w, feat.astype(np.float64))
        # projected first: P_l = W0_l . x_l at the level's own resolution, then resize and sample the 512 channels
        P = np.einsum("nc,bchw->bnhw", w, x.astype(np.float64)).astype(np.float32)
        got = O.grid_sample_2d(O.resize_bilinear_align_corners(P, O.MAP_SIZE), grid).astype(np.float64)
        scale = np.abs(ref).max()
        assert scale > 1e-3
        assert np.abs(got - ref).max() < 5e-6 * max(scale, 1.0), (C, float(np.abs(got - ref).max()), float(scale))
        col += C
    assert col == F - 3 and (B, N) == (q.shape[0], q.shape[1])


def test_rule_for_the_projected_levels():
    from list_amd import hip

    class T:                                              # stands in for a tensor: only .shape is read
        def __init__(self, *s):
            self.shape = s
    pyr = lambda r: [T(1, c, max(r >> s, 1), max(r >> s, 1)) for s, c in enumerate((64, 64, 128, 256, 512))]
    assert hip.img_proj_kept_levels(pyr(224), 137) == 2   # 56^2, 28^2, 14^2 are enlarged at least 2 x 2 by the resize
    assert hip.img_proj_kept_levels(pyr(512), 274) == 2   # BASELINE config 5: 128^2, 64^2, 32^2
    assert hip.img_proj_kept_levels(pyr(64), 137) == 0    # the small golden cases: every level
    assert hip.img_proj_kept_levels(pyr(1024), 137) == 4  # only the coarsest level is small enough
    assert hip.img_proj_kept_levels(pyr(4096), 137) == 5  # nothing to project: prep_img_proj refuses, hotpath keeps the standard map
