"""GPU parity tests of the backward: list_sdf_query_bwd through the C ABI against the
gradients autograd produced THROUGH THE REFERENCE's modules (tests/golden/hotpath_grad_*.npz) and against
the oracle's autograd restatement.

ReLU masks are discontinuous: a pre-activation within arithmetic noise of zero flips its mask and moves
every gradient by percents.  The gradient cases (oracle/cases.py: gtiny, gsmall, gedge) are therefore
seeded so that no pre-activation of the fp32 forward is closer to zero than 4e-6, ten times the
bf16x3 path's error; there `bf16x3` must match the reference to 2e-4 of each tensor's largest entry
(measured 1.5e-5).  fp16 / bf16 operands flip ~1e-3 / ~1e-2 of the masks by construction, so for them the
stated bound is on the relative L2 error of every gradient tensor (measured 2-4 % / 6-12 %)."""
import os

import numpy as np
import pytest
import torch

from oracle import cases, synth, torch_ops as TO
from test_oracle_golden import slice_like_golden

pytestmark = pytest.mark.gpu

TOL_X3_RELMAX = 2e-4
TOL_L2 = {"fp16": 0.08, "bf16": 0.25}


@pytest.fixture(scope="module")
def hip():
    import __graft_entry__ as ge
    ge.build()
    from list_amd import hip as h
    h.load()
    assert torch.cuda.is_available(), "the gpu-marked tests need a GPU"
    return h


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


def hip_gradients(hip, c, grad_sdf, precision, sort_points=True, want=None, map_size=137):
    md = hip.map_dtype_for(precision)
    img_in = [dev(m) for m in c["img_maps"]]
    img = hip.prep_img_maps(img_in, map_size, dtype=md)
    vox = hip.prep_vox_maps([dev(m) for m in c["vox_maps"]], dtype=md)
    params = {k: dev(v) for k, v in c["weights"].items()}
    packed = hip.prep_mlp_weights(params, vox.channels, img.channels, precision)
    packed_b = hip.prep_mlp_weights_bwd(params, vox.channels, img.channels, precision)
    sdf, ctx = hip.sdf_query(dev(c["query"]), dev(c["trans_mat"]), img, vox, packed, precision=precision,
                             save_for_backward=True, sort_points=sort_points)
    want = dict(want or {})
    in_call = want.pop("levels_in_call", False)
    out = hip.sdf_query_backward(ctx, dev(grad_sdf), packed_b, img_levels_like=img_in if in_call else None, **want)
    got = {}
    if "trans_mat" in out:
        got["d_trans_mat"] = out["trans_mat"]
    if "mlp" in out:
        got.update({"d_" + k: v for k, v in out["mlp"].items()})
    if "vox" in out:
        got.update({f"d_vox{i}": v.permute(0, 4, 1, 2, 3) for i, v in enumerate(out["vox"])})
    if "img_levels" in out:
        got.update({f"d_img{i}": v for i, v in enumerate(out["img_levels"])})
        if "img_map" in out:                 # (not with the half-precision intermediate, want_img_map=False)
            got["img_map"] = out["img_map"]
    elif "img_map" in out:
        got.update({f"d_img{i}": v for i, v in enumerate(hip.img_map_grad_to_levels(out["img_map"], img_in))})
        got["img_map"] = out["img_map"]
    torch.cuda.synchronize()
    return sdf, {k: v.cpu().numpy() for k, v in got.items()}


def rel_max(a, ref):
    return float(np.abs(a - ref).max()) / max(float(np.abs(ref).max()), 1e-30)


def rel_l2(a, ref):
    return float(np.linalg.norm((a - ref).ravel())) / max(float(np.linalg.norm(ref.ravel())), 1e-30)


# ------------------------------------------------------------------------------------------ kernels
@pytest.mark.parametrize("P,M,N", [(256, 256, 256), (512, 256, 264), (1024, 512, 3648), (4096, 256, 512)])
def test_transposed_gemm_matches_fp64(hip, P, M, N):
    """k_gemm_tn (ds_read_b64_tr_b16 operands): out = A^T B, contraction over the row index of both."""
    a = synth.normalish(1, (P, M))
    b = synth.uniform(2, (P, N), -1, 1)
    ref = a.astype(np.float64).T @ b.astype(np.float64)
    scale = np.abs(a).astype(np.float64).T @ np.abs(b).astype(np.float64)
    for prec, tol in (("bf16x3", 3e-5), ("fp16", 1.5e-3), ("bf16", 1e-2)):
        out = hip.gemm_tn(dev(a), dev(b), prec).cpu().numpy()
        assert (np.abs(out - ref) / scale).max() < tol, prec


def test_transposed_gemm_identity_asymmetric(hip):
    """A = [I] picks rows of an asymmetric B: catches a transposed store or a wrong point order."""
    a = np.eye(256, 256, dtype=np.float32)
    b = (np.arange(256 * 256, dtype=np.float32).reshape(256, 256) % 251) / 16.0
    out = hip.gemm_tn(dev(a), dev(b), "bf16x3").cpu().numpy()
    np.testing.assert_array_equal(out, b)
    out2 = hip.gemm_tn(dev(b), dev(a), "bf16x3").cpu().numpy()
    np.testing.assert_array_equal(out2, b.T)


def test_adjoint_resize_is_the_transpose_of_the_resize(hip):
    """<resize(x), g> == <x, adjoint(g)> for every level, on the forward's own kernel."""
    B, ms = 2, 137
    shapes = synth.img_map_shapes(B, 64)
    xs = [dev(synth.normalish(10 + i, s)) for i, s in enumerate(shapes)]
    Ct = sum(s[1] for s in shapes)
    g = dev(synth.normalish(20, (B, ms, ms, Ct)))
    y = hip.prep_img_maps(xs, ms, "f32").data
    adj = hip.img_map_grad_to_levels(g, xs)
    lhs = float((y.double() * g.double()).sum())
    rhs = float(sum((x.double() * a.double()).sum() for x, a in zip(xs, adj)))
    assert abs(lhs - rhs) <= 1e-5 * max(abs(lhs), 1.0)
    # channels-last destinations get the same values
    xs_cl = [x.contiguous(memory_format=torch.channels_last) for x in xs]
    adj_cl = hip.img_map_grad_to_levels(g, xs_cl)
    for a, b in zip(adj, adj_cl):
        assert b.is_contiguous(memory_format=torch.channels_last)
        assert torch.equal(a, b)


# ------------------------------------------------------------------------------------------ the path
@pytest.mark.parametrize("name", cases.GRAD_CASE_NAMES)
def test_backward_matches_reference_autograd(hip, golden_dir, name):
    g = np.load(os.path.join(golden_dir, f"hotpath_grad_{name}.npz"))
    c = cases.build_case(name)
    _, got = hip_gradients(hip, c, g["grad_sdf"], "bf16x3")
    keys = [k for k in g.files if k.startswith("d_")]
    assert len(keys) == 20
    for k in keys:
        a = slice_like_golden(name, k, got[k])
        assert a.shape == g[k].shape, k
        assert np.isfinite(a).all(), k
        assert rel_max(a, g[k]) < TOL_X3_RELMAX, (k, rel_max(a, g[k]))


@pytest.mark.parametrize("precision", ["fp16", "bf16"])
@pytest.mark.parametrize("name", cases.GRAD_CASE_NAMES)
def test_backward_reduced_precision_is_within_the_stated_l2_bound(hip, golden_dir, name, precision):
    g = np.load(os.path.join(golden_dir, f"hotpath_grad_{name}.npz"))
    _, got = hip_gradients(hip, cases.build_case(name), g["grad_sdf"], precision)
    for k in [k for k in g.files if k.startswith("d_")]:
        a = slice_like_golden(name, k, got[k])
        assert np.isfinite(a).all(), k
        assert rel_l2(a, g[k]) < TOL_L2[precision], (k, rel_l2(a, g[k]))
    # d fc_out.{weight,bias} do not depend on any mask: tight even here
    assert rel_max(got["d_fc_out.bias"], g["d_fc_out.bias"]) < 1e-5
    assert rel_max(got["d_fc_out.weight"], g["d_fc_out.weight"]) < (2e-3 if precision == "fp16" else 3e-2)      # fp16: 6e-4 measured


def test_backward_is_independent_of_point_order_and_partial_outputs(hip, golden_dir):
    """no_sort (atomic fallback of the perceptual-map gradient, row order for everything else) and
    requests for a subset of the outputs give the same gradients."""
    name = "gtiny"
    g = np.load(os.path.join(golden_dir, f"hotpath_grad_{name}.npz"))
    c = cases.build_case(name)
    _, a = hip_gradients(hip, c, g["grad_sdf"], "bf16x3")
    _, b = hip_gradients(hip, c, g["grad_sdf"], "bf16x3", sort_points=False)
    for k in a:
        assert rel_max(b[k], a[k]) < 1e-5, k
    _, only_mlp = hip_gradients(hip, c, g["grad_sdf"], "bf16x3",
                                want=dict(want_img=False, want_vox=False, want_trans=False))
    assert set(only_mlp) == {"d_" + k for k in c["weights"]}
    for k in only_mlp:
        assert rel_max(only_mlp[k], a[k]) < 1e-5, k
    _, inline = hip_gradients(hip, c, g["grad_sdf"], "bf16x3", want=dict(overlap=False))   # no auxiliary streams
    for k in a:
        assert rel_max(inline[k], a[k]) < 1e-5, k
    _, only_maps = hip_gradients(hip, c, g["grad_sdf"], "bf16x3", want=dict(want_mlp=False))
    for k in only_maps:
        assert rel_max(only_maps[k], a[k]) < 1e-5, k


def test_backward_on_the_real_map_sizes_against_the_oracle(hip):
    """137^2 x 1024 map from 224^2 encoder levels (down- AND up-sampled levels), 128^3 voxel levels:
    oracle autograd on the CPU, a few points."""
    c = cases.build_case("real")
    gs = synth.normalish(77, c["query"].shape[:2])
    args = TO.to_torch(c)
    _, ref = TO.list_query_grads(*args, torch.from_numpy(gs))
    _, got = hip_gradients(hip, c, gs, "bf16x3")
    bad = {}
    for k, r in ref.items():
        e = rel_max(got[k], r.numpy())
        if e > 5e-2:
            bad[k] = e
        assert np.isfinite(got[k]).all()
    # 64 points, masks may flip (the "real" case is not margin-seeded): the voxel/perceptual scatters and
    # the resize adjoint are checked through tensors that a single flip perturbs by << 5 %
    assert not bad, bad


@pytest.mark.parametrize("name", ["gsmall", "gtiny"])
def test_voxel_adjoint_forms_agree_with_the_reference(hip, golden_dir, name):
    """The three ways a voxel level's gradient is formed (global atomics, LDS windows, voxel-side gather)
    are chosen per level by size; forcing the gather everywhere / nowhere must give the reference's values."""
    g = np.load(os.path.join(golden_dir, f"hotpath_grad_{name}.npz"))
    c = cases.build_case(name)
    for mode in ("gather", "scatter"):
        _, got = hip_gradients(hip, c, g["grad_sdf"], "bf16x3",
                               want=dict(want_mlp=False, want_img=False, want_trans=False, vox_adjoint=mode))
        for i in range(6):
            k = f"d_vox{i}"
            a = slice_like_golden(name, k, got[k])
            assert rel_max(a, g[k]) < TOL_X3_RELMAX, (mode, k, rel_max(a, g[k]))


def test_packed_half_atomics_agree_with_the_fp32_atomics(hip):
    """fp16 operands: a sparse C = 32 level whose fp16 image fits the scratch, and the LDS-window levels, take packed-half atomics
    (global_atomic_pk_add_f16 into an image kept at the gradient scale, then one pass to fp32); forcing the direct
    form (vox_adjoint='scatter') keeps fp32 atomics on the same dX.  A voxel sums a handful of contributions, each
    rounded to 11 bits: the two agree to ~1e-3 of the level's largest entry and are not the same bits."""
    c = cases._case(seed=515, batch=2, n=500, img_res=64, vox_res=32)      # level 2: 16^3 x 32, 0.85 samples per cell
    gs = synth.normalish(9, (2, 500))
    _, a = hip_gradients(hip, c, gs, "fp16", want=dict(want_mlp=False, want_img=False, want_trans=False))
    _, b = hip_gradients(hip, c, gs, "fp16", want=dict(want_mlp=False, want_img=False, want_trans=False,
                                                       vox_adjoint="scatter"))
    assert np.isfinite(a["d_vox2"]).all()
    assert rel_max(a["d_vox2"], b["d_vox2"]) < 2e-3
    assert not np.array_equal(a["d_vox2"], b["d_vox2"])                    # the packed form did run
    _, ref = hip_gradients(hip, c, gs, "bf16x3", want=dict(want_mlp=False, want_img=False, want_trans=False))
    assert rel_l2(a["d_vox2"], ref["d_vox2"]) < TOL_L2["fp16"]
    for k in ("d_vox0", "d_vox1"):                                          # fp32 atomics either way
        assert rel_max(a[k], b[k]) < 1e-4, k
    # the LDS-window levels flush as packed halfs too (tens of flush-adds per voxel, each rounded to 11 bits)
    for k in ("d_vox3", "d_vox4", "d_vox5"):
        assert np.isfinite(a[k]).all(), k
        assert rel_max(a[k], b[k]) < 5e-3, (k, rel_max(a[k], b[k]))
        assert rel_l2(a[k], ref[k]) < TOL_L2["fp16"], k


def test_backward_large_batch_statistics(hip):
    """B = 8 x 6000 points (several Morton runs per voxel, many workgroups per image, rows padded): the
    HIP gradients in fp16 against the HIP gradients in bf16x3 (same kernels, 2-4 % mask-flip noise), and
    linearity in d(sdf): backward(2 g) == 2 backward(g) for the scale-free bf16x3 path."""
    c = cases._case(seed=909, batch=4, n=6000, img_res=64, vox_res=32)
    gs = synth.normalish(5, (4, 6000))
    _, a = hip_gradients(hip, c, gs, "bf16x3")
    _, a2 = hip_gradients(hip, c, 2.0 * gs, "bf16x3")
    # (not bitwise: the Morton order inside a sort bin, hence the fp32 summation order, varies run to run)
    for k in ("d_fc_0.weight", "d_fc_1.bias", "d_fc_out.weight", "d_vox3", "d_img2", "d_trans_mat"):
        assert rel_max(a2[k], 2.0 * a[k]) < 2e-5, k
    _, h = hip_gradients(hip, c, gs, "fp16")
    for k in a:
        assert rel_l2(h[k], a[k]) < TOL_L2["fp16"], (k, rel_l2(h[k], a[k]))


def test_trans_mat_gradient_is_stable_beside_the_weight_gradient(hip):
    """Regression (round 4): in the forked backward k_trans_grad used to run while dW0 did, and whenever one of its waves shared a
    SIMD with the weight-gradient GEMM's one point's v-derivative came out different: d_trans_mat off by 1e-4 .. 7e-3 of its
    largest entry in 7 - 100 % of the calls once other work had moved the allocator, every other gradient bit-stable
    (localised to one ds_bpermute pair of the kernel's loop, which is gone; list_capi.hip orders the stage behind dW0 as well).
    The order of the fp32 atomics alone moves d_trans_mat by 3e-7."""
    c = cases._case(seed=909, batch=4, n=6000, img_res=64, vox_res=32)
    gs = synth.normalish(5, (4, 6000))
    big = cases._case(seed=8181, batch=2, n=1500, img_res=64, vox_res=128)
    want = dict(want_mlp=True, want_img=False, want_vox=True, want_trans=True)
    outs = []
    for rnd in range(2):
        hip_gradients(hip, big, synth.normalish(1, (2, 1500)), "fp16")             # other shapes in between: the buffers move
        for _ in range(12):
            outs.append(hip_gradients(hip, c, gs, "bf16x3", want=want)[1]["d_trans_mat"].astype(np.float64))
    med = np.median(np.stack(outs), axis=0)
    worst = max(float(np.abs(o - med).max()) for o in outs) / float(np.abs(med).max())
    assert worst < 5e-6, worst


def relu_margin(c, map_size=137):
    """Smallest |pre-activation| of the fp32 forward relative to its layer's median (oracle, CPU)."""
    q, im, vx, T, W = TO.to_torch(c)
    with torch.no_grad():
        pts = q[:, :, [2, 1, 0]] * 2
        h = torch.cat((TO.stencil_voxel_features(pts, vx), TO.pooled_image_features(im, pts, T, map_size),
                       pts.transpose(1, 2)), 1)
        low = float("inf")
        for name in ("fc_0", "fc_1", "fc_2"):
            z = torch.nn.functional.conv1d(h, W[name + ".weight"], W[name + ".bias"])
            low = min(low, float(z.abs().min()) / float(z.abs().median()))
            h = torch.relu(z)
    return low


def margin_case(make, seeds):
    for seed in seeds:
        c = make(seed)
        if relu_margin(c, getattr(make, "map_size", 137)) > 1.5e-5:
            return c
    pytest.skip("no seed with a ReLU margin found")


def compare_with_oracle(hip, c, map_size=137, **kw):
    gs = synth.normalish(31, c["query"].shape[:2])
    _, ref = TO.list_query_grads(*TO.to_torch(c), torch.from_numpy(gs), map_size=map_size)
    _, got = hip_gradients(hip, c, gs, "bf16x3", map_size=map_size, **kw)
    for k, r in ref.items():
        assert np.isfinite(got[k]).all(), k
        assert rel_max(got[k], r.numpy()) < TOL_X3_RELMAX, (k, rel_max(got[k], r.numpy()))


def test_another_architecture_forward_and_backward(hip):
    """The ABI is not tied to the default widths: six voxel levels of 1/8/16/32/64/256 channels (im_enc_layers =
    [.., 8, 16, 32, 64, 256]) and a narrower 2-D pyramid (32/32/64/128/256 = 512 channels): F = 377*7 + 512 + 3 = 3154."""
    vox_c, img_c = (1, 8, 16, 32, 64, 256), (32, 32, 64, 128, 256)

    def make(seed):
        B, n, vox_res, img_res = 2, 77, 32, 64
        res = [vox_res, vox_res, vox_res // 2, vox_res // 4, vox_res // 8, vox_res // 16]
        vox = [synth.uniform(seed + 1000, (B, 1, res[0], res[0], res[0]))]
        vox += [synth.normalish(seed + 1000 + 13 * i, (B, c, r, r, r)) for i, (c, r) in enumerate(zip(vox_c[1:], res[1:]), 1)]
        img = [synth.normalish(seed + 11 * i, (B, c, max(img_res >> i, 1), max(img_res >> i, 1))) for i, c in enumerate(img_c)]
        F = sum(vox_c) * 7 + sum(img_c) + 3
        w = synth.make_mlp_weights(seed, feature_size=F, h_dim=256)
        return {"query": synth.make_query(seed, B, n), "img_maps": img, "vox_maps": vox,
                "trans_mat": synth.make_trans_mat(seed, B), "weights": w}
    c = margin_case(make, range(700, 760))
    assert c["weights"]["fc_0.weight"].shape[1] == 3154
    ref_sdf, _ = TO.list_query_grads(*TO.to_torch(c), torch.zeros(c["query"].shape[:2]))
    sdf, _ = hip_gradients(hip, c, np.zeros(c["query"].shape[:2], np.float32), "bf16x3")
    assert float((sdf.cpu() - ref_sdf).abs().max()) < 1e-4
    compare_with_oracle(hip, c)
    compare_with_oracle(hip, c, want=dict(vox_adjoint="scatter"))


def test_backward_on_odd_shapes(hip):
    """Non-square odd 2-D levels and non-cubic voxel levels: the adjoint resize, the three voxel adjoint forms' level
    classification and the image-gradient gather on sizes that fit no fast path."""
    img_hw = [(50, 46), (25, 23), (13, 12), (7, 6), (4, 3)]
    vox_dhw = [(12, 20, 28), (12, 20, 28), (6, 10, 14), (3, 5, 7), (2, 3, 4), (1, 2, 2)]

    def make(seed):
        B, n = 2, 61
        img = [synth.normalish(seed + i, (B, c, h, w)) for i, (c, (h, w)) in enumerate(zip(synth.IMG_CHANNELS, img_hw))]
        vox = [synth.uniform(seed + 50, (B, 1) + vox_dhw[0])]
        vox += [synth.normalish(seed + 50 + i, (B, c) + d) for i, (c, d) in enumerate(zip(synth.VOX_CHANNELS[1:], vox_dhw[1:]), 1)]
        return {"query": synth.make_query(seed, B, n), "img_maps": img, "vox_maps": vox,
                "trans_mat": synth.make_trans_mat(seed, B), "weights": synth.make_mlp_weights(seed)}
    c = margin_case(make, range(800, 860))
    compare_with_oracle(hip, c)
    compare_with_oracle(hip, c, want=dict(vox_adjoint="gather"))
    compare_with_oracle(hip, c, want=dict(overlap=False, levels_in_call=True))


def test_backward_with_a_map_wider_than_the_pixel_sort(hip):
    """map_size = 190 (like config 5's 274): the forward builds no pixel order, so the perceptual-map
    gradient takes the atomic form and trans_mat / the resize adjoint run at another size."""
    def make(seed):
        return cases._case(seed=seed, batch=2, n=40, img_res=64, vox_res=16)
    make.map_size = 190
    compare_with_oracle(hip, margin_case(make, range(3100, 3160)), map_size=190)


def test_backward_with_more_images_than_sort_slots(hip):
    """B = 70 > 64 image slots of the point sort (slots alias modulo 64): runs that mix images fall back
    to the direct forms; every image's gradients must still be its own."""
    def make(seed):
        return cases._case(seed=seed, batch=70, n=5, img_res=32, vox_res=16)
    # (350 points x 1024 units: seed 3249 has the widest ReLU margin of 3200..3299, 2.0e-5 of the layer median)
    compare_with_oracle(hip, margin_case(make, range(3249, 3300)))


@pytest.mark.parametrize("precision", ["bf16x3", "fp16"])
def test_zero_upstream_gradient_gives_exact_zeros(hip, precision):
    """d(loss)/d(sdf) = 0 (e.g. a masked batch): every output is written, finite and exactly zero (the fp16
    gradient scale falls back to 1)."""
    c = cases.build_case("gtiny")
    _, got = hip_gradients(hip, c, np.zeros(c["query"].shape[:2], np.float32), precision)
    assert len(got) == 21
    for k, v in got.items():
        assert np.isfinite(v).all() and not v.any(), k


# ------------------------------------------------------------------------------------------ errors
@pytest.mark.gpu
@pytest.mark.parametrize("overlap", [True, False])
def test_adjoint_resize_inside_the_call_equals_the_separate_call(hip, golden_dir, overlap):
    """ListQueryGradArgs.grad_img_levels: the same kernel on the same map gradient, only issued inside
    list_sdf_query_bwd (beside the scatters still in flight) -> identical bits; the other outputs unchanged."""
    name = "gsmall"
    g = np.load(os.path.join(golden_dir, f"hotpath_grad_{name}.npz"))
    c = cases.build_case(name)
    _, a = hip_gradients(hip, c, g["grad_sdf"], "bf16x3", want=dict(overlap=overlap))
    _, b = hip_gradients(hip, c, g["grad_sdf"], "bf16x3", want=dict(overlap=overlap, levels_in_call=True))
    assert set(a) == set(b)
    for i in range(5):
        from_map = hip.img_map_grad_to_levels(dev(b["img_map"]), [dev(m) for m in c["img_maps"]])[i].cpu().numpy()
        np.testing.assert_array_equal(b[f"d_img{i}"], from_map)
    for k in a:
        assert rel_max(b[k], a[k]) < 1e-5, k
    # the level gradients need the map gradient as their intermediate
    with pytest.raises(RuntimeError, match="channel counts"):
        img_in = [dev(m) for m in c["img_maps"]]
        img = hip.prep_img_maps(img_in, 137, dtype="f32")
        vox = hip.prep_vox_maps([dev(m) for m in c["vox_maps"]], dtype="f32")
        params = {k: dev(v) for k, v in c["weights"].items()}
        packed = hip.prep_mlp_weights(params, vox.channels, img.channels, "bf16x3")
        packed_b = hip.prep_mlp_weights_bwd(params, vox.channels, img.channels, "bf16x3")
        sdf, ctx = hip.sdf_query(dev(c["query"]), dev(c["trans_mat"]), img, vox, packed, save_for_backward=True)
        hip.sdf_query_backward(ctx, torch.zeros_like(sdf), packed_b, img_levels_like=img_in[:4] + [img_in[4][:, :8]])


def test_half_precision_map_gradient_between_gather_and_adjoint_resize(hip, golden_dir):
    """ABI 6, grad_img_map_dtype = F16 (fp16 operands, levels in the call, want_img_map=False): the map gradient between
    the map-side gather and the adjoint resize travels as halfs at the gradient scale.  Same level gradients up to the
    11-bit rounding of the intermediate (2^-11 per map pixel, averaged down by the resize's footprints); nothing else
    moves; the fp32-grade mode and the forms without the pixel order keep the fp32 intermediate."""
    g = np.load(os.path.join(golden_dir, "hotpath_grad_gsmall.npz"))
    c = cases.build_case("gsmall")
    _, a = hip_gradients(hip, c, g["grad_sdf"], "fp16", want={"levels_in_call": True})
    _, b = hip_gradients(hip, c, g["grad_sdf"], "fp16", want={"levels_in_call": True, "want_img_map": False})
    assert "img_map" in a and "img_map" not in b
    for i in range(5):
        ref = a[f"d_img{i}"]
        assert np.isfinite(b[f"d_img{i}"]).all()
        assert rel_max(b[f"d_img{i}"], ref) < 1.5e-3, (i, rel_max(b[f"d_img{i}"], ref))
        assert rel_l2(b[f"d_img{i}"], ref) < 5e-4, (i, rel_l2(b[f"d_img{i}"], ref))
    for k in a:
        if not k.startswith("d_img") and k != "img_map":
            if k.startswith("d_vox") or k == "d_trans_mat":        # atomically summed (packed halfs on some levels):
                assert rel_max(b[k], a[k]) < 5e-3, k              # they vary run to run by 1-2e-3
            else:
                np.testing.assert_array_equal(b[k], a[k], err_msg=k)
    # bf16x3 never takes the half intermediate; neither does an unsorted query
    _, x3 = hip_gradients(hip, c, g["grad_sdf"], "bf16x3", want={"levels_in_call": True, "want_img_map": False})
    assert "img_map" in x3
    _, un = hip_gradients(hip, c, g["grad_sdf"], "fp16", sort_points=False, want={"levels_in_call": True, "want_img_map": False})
    assert "img_map" in un
    for i in range(5):
        assert rel_l2(un[f"d_img{i}"], a[f"d_img{i}"]) < 1e-5 + 0.05        # (different summation order and atomics: loose)


def test_backward_rejects_unsupported_calls(hip):
    c = cases.build_case("gtiny")
    md = "f32"
    img = hip.prep_img_maps([dev(m) for m in c["img_maps"]], dtype=md)
    vox = hip.prep_vox_maps([dev(m) for m in c["vox_maps"]], dtype=md)
    params = {k: dev(v) for k, v in c["weights"].items()}
    packed = hip.prep_mlp_weights(params, vox.channels, img.channels, "bf16x3")
    packed_b = hip.prep_mlp_weights_bwd(params, vox.channels, img.channels, "bf16x3")
    q = dev(c["query"])
    sdf, ctx = hip.sdf_query(q, dev(c["trans_mat"]), img, vox, packed, save_for_backward=True)
    with pytest.raises(RuntimeError, match="float32 CUDA"):
        hip.sdf_query_backward(ctx, torch.zeros(sdf.shape), packed_b)
    # a forward workspace that cannot hold the query in one chunk is refused, not misread
    keep = ctx.args.workspace_bytes
    ctx.args.workspace_bytes = 1 << 20
    with pytest.raises(RuntimeError, match="one chunk|workspace"):
        hip.sdf_query_backward(ctx, torch.zeros_like(sdf), packed_b)
    # a forward that was told to keep no activations (inference: fc_1 + fc_2 + fc_out as one kernel where the
    # operands allow it) left no H1 / H2 behind: the backward refuses it instead of reading stale planes
    ctx.args.workspace_bytes = keep
    ctx.args.no_activations = 1
    with pytest.raises(hip.ListError, match="no_activations") as e:
        hip.sdf_query_backward(ctx, torch.zeros_like(sdf), packed_b)
    assert e.value.code == hip.ERR_ARG


# ------------------------------------------------------------------------------------------ autograd
def test_autograd_function_routes_through_the_hip_backward(hip, golden_dir, monkeypatch):
    """network.hotpath.sdf_query(...).backward() == the golden gradients; there is no torch-op
    re-evaluation path left in the package."""
    from list_amd.network import hotpath
    name = "gsmall"
    g = np.load(os.path.join(golden_dir, f"hotpath_grad_{name}.npz"))
    c = cases.build_case(name)
    assert not hasattr(hotpath, "_recompute_with_torch_ops")
    leaf = lambda a: dev(a).requires_grad_(True)
    img = [leaf(m) for m in c["img_maps"]]
    vox = [leaf(m) for m in c["vox_maps"]]
    T = leaf(c["trans_mat"])
    W = {k: leaf(v) for k, v in c["weights"].items()}
    sdf = hotpath.sdf_query(dev(c["query"]), T, img, vox, W, precision="bf16x3")
    assert sdf.requires_grad
    (sdf * dev(g["grad_sdf"])).sum().backward()
    got = {"d_trans_mat": T.grad}
    got.update({f"d_img{i}": t.grad for i, t in enumerate(img)})
    got.update({f"d_vox{i}": t.grad for i, t in enumerate(vox)})
    got.update({"d_" + k: t.grad for k, t in W.items()})
    for k in [k for k in g.files if k.startswith("d_")]:
        a = slice_like_golden(name, k, got[k].cpu().numpy())
        assert a.shape == g[k].shape, k
        assert rel_max(a, g[k]) < TOL_X3_RELMAX, (k, rel_max(a, g[k]))
    # only some inputs need gradients
    T2 = leaf(c["trans_mat"])
    sdf2 = hotpath.sdf_query(dev(c["query"]), T2, [t.detach() for t in img], [t.detach() for t in vox],
                             {k: t.detach() for k, t in W.items()}, precision="bf16x3")
    (sdf2 * dev(g["grad_sdf"])).sum().backward()
    assert rel_max(T2.grad.cpu().numpy(), g["d_trans_mat"]) < TOL_X3_RELMAX


def test_half_precision_voxel_leaves_get_half_gradients(hip):
    """A half-precision 3-D encoder: fp16 channels-last voxel leaves are used where they lie; their
    gradients come back in fp16 and equal the fp32 leaves' gradients (same values in the maps) after rounding."""
    from list_amd.network import hotpath
    c = cases.build_case("gtiny")
    gsdf = torch.randn((c["query"].shape[0], c["query"].shape[1]), device="cuda:0")
    W = {k: dev(v) for k, v in c["weights"].items()}
    img = [dev(m) for m in c["img_maps"]]
    grads = {}
    for kind in ("half", "float"):
        vox = []
        for m in c["vox_maps"]:
            t = dev(m).half()
            t = t.contiguous(memory_format=torch.channels_last_3d) if kind == "half" else t.float()
            vox.append(t.requires_grad_(True))
        sdf = hotpath.sdf_query(dev(c["query"]), dev(c["trans_mat"]), img, vox, W, precision="fp16")
        (sdf * gsdf).sum().backward()
        grads[kind] = [t.grad for t in vox]
        grads[kind + "_sdf"] = sdf.detach()
    assert torch.equal(grads["half_sdf"], grads["float_sdf"])
    for h, f in zip(grads["half"], grads["float"]):
        assert h.dtype == torch.float16 and f.dtype == torch.float32 and h.shape == f.shape
        scale = float(f.abs().max()) + 1e-30
        assert float((h.float() - f).abs().max()) <= 2e-3 * scale          # fp16 rounding of the same sums (+ atomics order)


def _leaves(c):
    leaf = lambda a: dev(a).requires_grad_(True)
    return ([leaf(m) for m in c["img_maps"]], [leaf(m) for m in c["vox_maps"]], leaf(c["trans_mat"]),
            {k: leaf(v) for k, v in c["weights"].items()})


def test_long_queries_are_cut_into_pieces(hip, golden_dir, monkeypatch):
    """Above the per-call point limit the autograd function evaluates and differentiates the query in
    pieces along the point axis; gradients are the sums (limit lowered to 128 points here)."""
    from list_amd.network import hotpath
    monkeypatch.setattr(hotpath, "HIP_BACKWARD_MAX_POINTS", 128)
    name = "gtiny"                                       # 2 x 129 points -> 3 pieces of <= 64 points per image
    g = np.load(os.path.join(golden_dir, f"hotpath_grad_{name}.npz"))
    c = cases.build_case(name)
    img, vox, T, W = _leaves(c)
    sdf = hotpath.sdf_query(dev(c["query"]), T, img, vox, W, precision="bf16x3")
    ref_sdf = np.load(os.path.join(golden_dir, f"hotpath_grad_{name}.npz"))
    (sdf * dev(g["grad_sdf"])).sum().backward()
    got = {"d_trans_mat": T.grad}
    got.update({f"d_img{i}": t.grad for i, t in enumerate(img)})
    got.update({f"d_vox{i}": t.grad for i, t in enumerate(vox)})
    got.update({"d_" + k: t.grad for k, t in W.items()})
    for k in [k for k in g.files if k.startswith("d_")]:
        a = slice_like_golden(name, k, got[k].cpu().numpy())
        assert rel_max(a, g[k]) < TOL_X3_RELMAX, (k, rel_max(a, g[k]))


def test_module_forms_differentiate_through_hip(hip, golden_dir):
    """The reference's own call pattern (models.py:94-97): PerceptualPooling.forward, then
    VoxelDecoder2.forward(p, feat, percep_feat) -- both differentiable through the C ABI
    (list_percep_pool_bwd, list_sdf_query_bwd with grad_percep_feat); chained, they must reproduce the
    golden gradients of the fused path."""
    from list_amd.network import modules
    name = "gsmall"
    g = np.load(os.path.join(golden_dir, f"hotpath_grad_{name}.npz"))
    c = cases.build_case(name)
    img, vox, T, W = _leaves(c)
    pool = modules.PerceptualPooling()
    dec = modules.VoxelDecoder2(3610, 256).to("cuda:0")
    dec.load_state_dict({"fc." + k: dev(v) for k, v in c["weights"].items()})
    q = dev(c["query"])[:, :, [2, 1, 0]] * 2
    percep = pool(img, q, T)                                       # [B,1024,1,N]
    assert percep.requires_grad
    B, N = q.shape[:2]
    sdf = dec(q, vox, percep.reshape(B, -1, N))
    (sdf * dev(g["grad_sdf"])).sum().backward()
    got = {"d_trans_mat": T.grad}
    got.update({f"d_img{i}": t.grad for i, t in enumerate(img)})
    got.update({f"d_vox{i}": t.grad for i, t in enumerate(vox)})
    got.update({"d_" + k[3:]: p.grad for k, p in dec.named_parameters()})
    for k in [k for k in g.files if k.startswith("d_")]:
        a = slice_like_golden(name, k, got[k].cpu().numpy())
        assert a.shape == g[k].shape, k
        assert rel_max(a, g[k]) < TOL_X3_RELMAX, (k, rel_max(a, g[k]))


def test_image_gradient_with_projections_piled_onto_the_clamp(hip):
    """Thousands of points on one pixel (network/modules.py:43 clamps what leaves the map; an untrained spatial
    transformer sends ~90 % of the points there): the map-side gather cuts such a pixel group into chunks summed by
    separate workgroups (bwd_scatter_kernels.hip, kHeavyChunk = 1024 candidates).  Checked against (a) the atomic form of
    the same gradient, which the unsorted forward takes (no pixel order, no groups), and (b) the oracle's autograd."""
    seed, B, N = 7711, 2, 2600
    c = {"query": synth.make_query(seed, B, N), "img_maps": synth.make_img_maps(seed, B, 32),
         "vox_maps": synth.make_vox_maps(seed, B, 16), "weights": synth.make_mlp_weights(seed),
         "trans_mat": synth.make_trans_mat(seed, B)}
    T = c["trans_mat"].copy()
    T[0] = np.array([[0, 0, 0], [0, 0, 0], [0, 0, 0], [-3, 500, 1]], np.float32)       # image 0: ALL points on pixel (0, 136)
    T[1, :3, :2] *= 40.0                                                                  # image 1: most points on the borders
    c["trans_mat"] = T
    gs = synth.normalish(78, (B, N))
    want = dict(want_mlp=False, want_vox=False)
    _, piled = hip_gradients(hip, c, gs, "bf16x3", want=want)                              # pixel order: gather (+ heavy chunks)
    _, atomic = hip_gradients(hip, c, gs, "bf16x3", sort_points=False, want=want)          # no pixel order: atomics
    assert np.abs(atomic["img_map"]).max() > 0
    assert rel_max(piled["img_map"], atomic["img_map"]) < 2e-5
    assert rel_max(piled["d_trans_mat"], atomic["d_trans_mat"]) < 1e-4
    # all of image 0's gradient mass sits on the two map rows / columns its one pixel touches
    g0 = np.abs(piled["img_map"][0]).sum(-1)
    assert g0[136, 0] > 0 and g0[:135].sum() == 0 and g0[:, 2:].sum() == 0
    args = TO.to_torch(c)
    _, ref = TO.list_query_grads(*args, torch.from_numpy(gs))
    for i in range(5):
        assert rel_max(piled[f"d_img{i}"], ref[f"d_img{i}"].numpy()) < 5e-2, i            # (masks may flip: not margin-seeded)
