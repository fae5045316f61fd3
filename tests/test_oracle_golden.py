"""CPU: the oracle (numpy restatement + torch-op restatement) against the golden vectors that
oracle/gen_golden.py produced by running the reference's own modules."""
import os

import numpy as np
import pytest

from oracle import cases, list_oracle as O, torch_ops as TO

TOL_SDF = 2e-6          # fp32, K=3610 dot products, |sdf| ~ 0.06
TOL_FEAT = 2e-5         # features are O(1..4); resize+sample rounding


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, f"hotpath_{name}.npz"))


ALL_CASES = cases.CASE_NAMES + cases.NONFINITE_CASE_NAMES


@pytest.mark.parametrize("name", ALL_CASES)
def test_numpy_oracle_matches_reference(golden_dir, name):
    """(assert_allclose treats NaN == NaN and +-inf == +-inf: the non-finite case pins their positions too)"""
    g = _load(golden_dir, name)
    c = cases.build_case(name)
    q = O.permute_scale_query(c["query"])
    B, N, _ = q.shape
    st = cases.FEATURE_STRIDE.get(name, 4)
    with np.errstate(invalid="ignore", over="ignore"):
        percep = O.perceptual_pooling(c["img_maps"], q, c["trans_mat"])
        assert percep.shape == (B, 1024, 1, N)
        np.testing.assert_allclose(percep[:, :, :, ::st], g["percep_sub"], rtol=0, atol=TOL_FEAT)
        vf = O.vox_features(q, c["vox_maps"])
        assert vf.shape == (B, 2583, N)
        np.testing.assert_allclose(vf[:, :, ::st], g["voxfeat_sub"], rtol=0, atol=TOL_FEAT)
        sdf = O.voxel_decoder2(q, c["vox_maps"], percep.reshape(B, -1, N), c["weights"])
    np.testing.assert_allclose(sdf, g["sdf"], rtol=0, atol=TOL_SDF)
    if name in cases.NONFINITE_CASE_NAMES:
        assert np.isnan(g["sdf"]).any() and np.isfinite(g["sdf"]).any()
        assert np.isinf(g["voxfeat_sub"]).any() and np.isnan(g["voxfeat_sub"]).any()
    r0 = O.resize_bilinear_align_corners(c["img_maps"][0], 137)
    np.testing.assert_allclose(r0[:, ::8, ::3, ::3], g["resized0_sub"], rtol=0, atol=2e-6)


@pytest.mark.parametrize("name", ALL_CASES)
def test_torch_restatement_matches_reference(golden_dir, name):
    g = _load(golden_dir, name)
    sdf = TO.list_query(*TO.to_torch(cases.build_case(name))).numpy()
    np.testing.assert_allclose(sdf, g["sdf"], rtol=0, atol=TOL_SDF)


GRAD_CASES = cases.GRAD_CASE_NAMES


def slice_like_golden(name, key, arr):
    """The slicing oracle/gen_golden.py:grad_goldens applied to keep the fixtures small."""
    if key == "d_fc_0.weight":
        return arr[::8]
    if name == "gsmall" and key.startswith("d_vox"):
        return arr[:, :, ::2, ::2, ::2]
    if name == "gsmall" and key.startswith("d_img"):
        return arr[:, :, ::2, ::2]
    return arr


@pytest.mark.parametrize("name", GRAD_CASES)
def test_torch_restatement_gradients_match_reference(golden_dir, name):
    """Backward: autograd over the restated op sequence == autograd through the reference's
    own modules, for every differentiable input."""
    import torch
    g = np.load(os.path.join(golden_dir, f"hotpath_grad_{name}.npz"))
    args = TO.to_torch(cases.build_case(name))
    _, grads = TO.list_query_grads(*args, torch.from_numpy(g["grad_sdf"]))
    keys = [k for k in g.files if k.startswith("d_")]
    assert len(keys) == 1 + 5 + 6 + 8
    for k in keys:
        got = slice_like_golden(name, k, grads[k].numpy())
        ref = g[k]
        assert got.shape == ref.shape, k
        scale = max(float(np.abs(ref).max()), 1e-6)
        assert float(np.abs(got - ref).max()) <= 2e-5 * scale, k


def test_feature_order_is_channel_major_stencil_minor(golden_dir):
    """k = c*7 + j (modules.py:270-273): feature 7*c is the centre sample of channel c."""
    c = cases.build_case("tiny")
    q = O.permute_scale_query(c["query"])
    vf = O.vox_features(q, c["vox_maps"])
    centre = O.grid_sample_3d_border(c["vox_maps"][1], q)          # 16 channels of level 1
    np.testing.assert_array_equal(vf[:, 7 * 1:7 * 17:7, :], centre)


def test_aux_grid_loss_stencil(golden_dir):
    a = np.load(os.path.join(golden_dir, "aux.npz"))
    np.testing.assert_array_equal(O.create_grid_points_from_bounds(-0.5, 0.5, 8), a["grid8"])
    np.testing.assert_array_equal(O.stencil(), a["displacements"])
    loss = O.sdf_loss(a["loss_outputs"], a["loss_targets"], 2.0)
    for k, v in loss.items():
        np.testing.assert_allclose(v, a["loss_" + k], rtol=2e-6)


def test_synth_is_stable():
    """The generator is the contract between fixtures and tests: pin a few values."""
    from oracle import synth
    u = synth.uniform(7, (5,))
    n = synth.normalish(7, (5,))
    assert u.dtype == np.float32 and n.dtype == np.float32
    np.testing.assert_array_equal(u, synth.uniform(7, (5,)))
    big = synth.normalish(3, (200000,))
    assert abs(float(big.mean())) < 0.01 and abs(float(big.std()) - 1.0) < 0.01
    uu = synth.uniform(3, (200000,), -0.5, 0.5)
    assert uu.min() >= -0.5 and uu.max() < 0.5 and abs(float(uu.mean())) < 0.005
