"""GPU: the nn.Module mirror (LIST / PerceptualPooling / VoxelDecoder2 / executor / train.py) running
the HIP path, against goldens from the reference's own LIST.forward and against torch on the device."""
import os

import numpy as np
import pytest
import torch

from oracle import fill, synth, torch_ops as TO
from list_amd import arguments, utils

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module", autouse=True)
def _built_library():
    import __graft_entry__ as ge
    ge.build()                      # no-op when csrc/liblist_hip.so is up to date


@pytest.fixture(scope="module")
def cfg():
    return arguments.default_config(vox_res=32, train_batch_size=2)


@pytest.fixture(scope="module")
def net(cfg):
    return fill.fill_state(utils.get_class("network.models.LIST")(cfg), seed=2).eval()


def test_list_forward_matches_reference_model(net, golden_dir):
    g = np.load(os.path.join(golden_dir, "models.npz"))
    img = torch.from_numpy(synth.uniform(78, (2, 3, 64, 64)))
    q = torch.from_numpy(synth.make_query(79, 2, 100))
    tm_given = torch.from_numpy(synth.make_trans_mat(80, 2))
    with torch.no_grad():
        # (a) per-image stage on the CPU (bit-comparable with the reference's CPU run), per-point stage on HIP
        feat_l2, vox_feat, tm, _, _ = net.encode(img)
        net_gpu_dec = net.sdf_decoder.to(DEV)
        sdf = net.query_sdf(q.to(DEV), [f.to(DEV) for f in feat_l2], [v.to(DEV) for v in vox_feat], tm.to(DEV))
        assert np.abs(sdf.cpu().numpy() - g["list_sdf"]).max() < 1e-4
        sdf2 = net.query_sdf(q.to(DEV), [f.to(DEV) for f in feat_l2], [v.to(DEV) for v in vox_feat],
                             tm_given.to(DEV))
        assert np.abs(sdf2.cpu().numpy() - g["list_sdf_given_transmat"]).max() < 1e-4
        # (b) everything on the GPU (MIOpen convolutions differ from the CPU's in the last bits)
        net.to(DEV)
        vox0, sdf3 = net(img.to(DEV), q.to(DEV))
        assert vox0.shape == (2, 1, 32, 32, 32) and sdf3.shape == (2, 100)
        err = np.abs(sdf3.cpu().numpy() - g["list_sdf"]).max()
        print(f"whole-GPU LIST.forward vs reference CPU: {err:.3e}")
        assert err < 2e-3
    net.cpu()


def test_module_level_calls_equal_fused(net, monkeypatch):
    """executors.LIST.test of the reference calls percep_pooling then sdf_decoder by attribute."""
    net.to(DEV)
    img = torch.from_numpy(synth.uniform(78, (2, 3, 64, 64))).to(DEV)
    q = torch.from_numpy(synth.make_query(5, 2, 333)).to(DEV)
    with torch.no_grad():
        feat_l2, vox_feat, tm, _, _ = net.encode(img)
        fused = net.query_sdf(q, feat_l2, vox_feat, tm)
        p = q[:, :, [2, 1, 0]] * 2
        percep = net.percep_pooling(feat_l2, p, tm)
        assert percep.shape == (2, 1024, 1, 333)
        two_step = net.sdf_decoder(p, vox_feat, percep.reshape(2, -1, 333))
        # an inference forward projects the low-resolution encoder levels through fc_0 before the resize (round 4b,
        # hip.prep_img_proj): the same field, other rounding (fp32-grade operands: a few 1e-7)
        assert float((fused - two_step).abs().max()) < 5e-6
        # without it: the same kernels and the same split of the same fp32 features -> identical
        monkeypatch.setenv("LIST_IMG_PROJ", "0")
        assert torch.equal(net.query_sdf(q, feat_l2, vox_feat, tm), two_step)
    net.cpu()


def test_encoders_emit_channels_last_maps_used_in_place(net):
    """the model's producers hand the query path channels-last maps, so the 3-D layout
    hand-off is zero-copy (the prepared levels alias the encoder outputs)."""
    from list_amd import hip
    net.to(DEV)
    img = torch.from_numpy(synth.uniform(78, (2, 3, 64, 64))).to(DEV)
    with torch.no_grad():
        feat_l2, vox_feat, tm, _, _ = net.encode(img)
    assert all(f.stride(1) == 1 for f in feat_l2)                       # NHWC image maps
    vox = hip.prep_vox_maps(vox_feat)
    for l, v in enumerate(vox_feat):
        if v.shape[1] > 1 and v.shape[2] > 1:
            assert v.is_contiguous(memory_format=torch.channels_last_3d)
            assert vox.levels[l].data == v.data_ptr()
    net.cpu()


def test_half_precision_voxel_encoder_hands_its_levels_over_in_place(cfg, net):
    """SURVEY 8 f2 ('optionally half precision'): --vox_encoder_precision fp16 runs the 3-D encoder under autocast;
    its fp16 channels-last levels are the fp16 maps of the query path (no copy), the SDF stays close to the fp32
    encoder's and a training step differentiates through it (fp16 voxel gradients into the encoder)."""
    from list_amd import hip
    cfg_h = arguments.default_config(vox_res=32, train_batch_size=2, precision="fp16", vox_encoder_precision="fp16")
    cfg_f = arguments.default_config(vox_res=32, train_batch_size=2, precision="fp16")
    half = utils.get_class("network.models.LIST")(cfg_h)
    full = utils.get_class("network.models.LIST")(cfg_f)
    half.load_state_dict(net.state_dict()), full.load_state_dict(net.state_dict())
    half.to(DEV).eval(), full.to(DEV).eval()
    img = torch.from_numpy(synth.uniform(78, (2, 3, 64, 64))).to(DEV)
    q = torch.from_numpy(synth.make_query(5, 2, 200)).to(DEV)
    with torch.no_grad():
        feat_l2, vox_feat, tm, _, _ = half.encode(img)
        assert all(v.dtype == torch.float16 for v in vox_feat)
        vox = hip.prep_vox_maps(vox_feat, "f16")
        for l, v in enumerate(vox_feat):
            if v.shape[1] > 1 and v.shape[2] > 1:
                assert vox.levels[l].data == v.data_ptr() and vox.levels[l].dtype == hip.MAP_F16
        occ_h, sdf_h = half(img, q)
        occ_f, sdf_f = full(img, q)
    assert occ_h.dtype == torch.float32 and torch.isfinite(sdf_h).all()
    assert float((sdf_h - sdf_f).abs().max()) < 5e-3 * max(1.0, float(sdf_f.abs().max()))
    half.train()
    occ, sdf = half(img, q)
    (sdf.square().mean() + occ.mean()).backward()
    g = half.vox_encoder.conv["conv_3"].weight.grad
    assert g is not None and g.dtype == torch.float32 and torch.isfinite(g).all() and float(g.abs().sum()) > 0


def test_prepared_map_cache_is_not_fooled_by_address_reuse(net):
    """Fresh encoder outputs usually land on the addresses of the freed previous ones (same shapes,
    version 0): the prepared-map cache must key on tensor identity, not on data_ptr."""
    net.to(DEV)
    q = torch.from_numpy(synth.make_query(9, 1, 300)).to(DEV)
    outs = []
    with torch.no_grad():
        for seed in (11, 12, 11):
            img = torch.from_numpy(synth.uniform(seed, (1, 3, 64, 64))).to(DEV)
            feat_l2, vox_feat, tm, _, _ = net.encode(img)
            outs.append(net.query_sdf(q, feat_l2, vox_feat, tm).clone())
            del feat_l2, vox_feat, tm, img           # let the allocator recycle the blocks
    assert not torch.equal(outs[0], outs[1])          # different image -> different field
    assert (outs[0] - outs[2]).abs().max() < 1e-5     # same image again -> same field (MIOpen noise only)
    net.cpu()


def _relu_margin(feats, weights):
    """Smallest |pre-activation| relative to the layer's median magnitude."""
    h, low = feats, float("inf")
    for name in ("fc_0", "fc_1", "fc_2"):
        z = torch.nn.functional.conv1d(h, weights[name + ".weight"], weights[name + ".bias"])
        low = min(low, float(z.abs().min()) / float(z.abs().median()))
        h = torch.relu(z)
    return low


def test_gradients_flow_through_the_query(net):
    """LIST.query_sdf(...).backward() runs list_sdf_query_bwd; compared with torch autograd on the device.
    ReLU masks are discontinuous, so the query seed is chosen such that no pre-activation is within
    1.5e-5 of its layer's median magnitude of zero (several times the arithmetic noise of bf16x3)."""
    net.to(DEV)
    img = torch.from_numpy(synth.uniform(78, (2, 3, 64, 64))).to(DEV)
    with torch.no_grad():
        feat_l2, vox_feat, tm, _, _ = net.encode(img)
        w0 = net.sdf_decoder.mlp_params()
        for seed in range(6, 60):
            q = torch.from_numpy(synth.make_query(seed, 2, 64)).to(DEV)
            pts0 = q[:, :, [2, 1, 0]] * 2
            f0 = torch.cat((TO.stencil_voxel_features(pts0, vox_feat), TO.pooled_image_features(feat_l2, pts0, tm),
                            pts0.transpose(1, 2)), 1)
            if _relu_margin(f0, w0) > 1.5e-5:
                break
        else:
            pytest.skip("no query seed with a ReLU margin found")
    leaves = [t.clone().requires_grad_(True) for t in (tm, feat_l2[0], vox_feat[3])]
    fl = [leaves[1]] + feat_l2[1:]
    vf = vox_feat[:3] + [leaves[2]] + vox_feat[4:]
    sdf = net.query_sdf(q, fl, vf, leaves[0])
    w = torch.linspace(-1, 1, sdf.numel(), device=DEV).view_as(sdf)
    (sdf * w).sum().backward()
    pw = net.sdf_decoder.fc["fc_0"].weight
    assert pw.grad is not None and torch.isfinite(pw.grad).all() and pw.grad.abs().sum() > 0
    # independent torch evaluation on the device
    ref_leaves = [t.detach().clone().requires_grad_(True) for t in leaves]
    weights = {k: v.detach().clone().requires_grad_(True) for k, v in net.sdf_decoder.mlp_params().items()}
    pts = q[:, :, [2, 1, 0]] * 2
    percep = TO.pooled_image_features([ref_leaves[1]] + [f.detach() for f in feat_l2[1:]], pts, ref_leaves[0])
    feats = torch.cat((TO.stencil_voxel_features(pts, [v.detach() for v in vox_feat[:3]] + [ref_leaves[2]]
                                                 + [v.detach() for v in vox_feat[4:]]), percep,
                       pts.transpose(1, 2)), 1)
    ref = TO.implicit_mlp(feats, weights)
    assert (ref - sdf).abs().max() < 1e-4
    (ref * w).sum().backward()
    def close(a, b):
        return float((a - b).abs().max()) <= 2e-4 * float(b.abs().max())
    for a, b in zip(leaves, ref_leaves):
        assert close(a.grad, b.grad)
    assert close(pw.grad, weights["fc_0.weight"].grad)
    net.zero_grad()
    net.cpu()


def test_executor_grid_prediction_on_the_projected_perceptual_map(cfg, net):
    """A grid with at least 4 x 137^2 points per image: predict_grid's queries take the projected perceptual map
    (fc_0's perceptual block applied to the map once per image, hip.prep_percep_proj) -- the same field as the
    standard per-point path up to rounding, in both operand formats."""
    net.to(DEV)
    from list_amd.train import _Module
    img = torch.from_numpy(synth.uniform(78, (1, 3, 64, 64))).to(DEV)
    with torch.no_grad():
        enc = net.encode(img)
    net.encode = lambda *a, **k: enc   # freeze the per-image stage so only the query path is compared
    prec0 = net.sdf_decoder.precision
    try:
        for prec, tol in (("bf16x3", 5e-6), ("fp16", 1e-4)):
            net.sdf_decoder.precision = prec
            cfg2 = arguments.default_config(vox_res=32, train_batch_size=2, mcube_znum=44, test_pointnum=5000)
            cfg2.device = torch.device(DEV)
            ex = utils.get_class("network.executors.LIST")(cfg2, _Module(net))
            vol, _, _ = ex.predict_grid(img)
            assert vol.shape == (44, 44, 44) and torch.isfinite(vol).all()
            assert "proj:" + prec in net.sdf_decoder._caches[("device", torch.device(DEV).index or 0)]
            grid = torch.tensor(utils.create_grid_points_from_bounds(-0.5, 0.5, 44)).float().unsqueeze(0).to(DEV)
            with torch.no_grad():
                one = net.query_sdf(grid, enc[0], enc[1], enc[2])          # standard path (sorted points, no projection)
            assert float((one.view(44, 44, 44) / cfg2.sdf_scale - vol).abs().max()) < tol
    finally:
        net.sdf_decoder.precision = prec0
        del net.encode
    net.cpu()


def test_executor_grid_prediction(cfg, net):
    """Inference driver (reference executors.py:191-231): chunked device-side grid == one-shot query."""
    net.to(DEV)
    from list_amd.train import _Module
    cfg2 = arguments.default_config(vox_res=32, train_batch_size=2, mcube_znum=24, test_pointnum=5000)
    cfg2.device = torch.device(DEV)
    ex = utils.get_class("network.executors.LIST")(cfg2, _Module(net))
    img = torch.from_numpy(synth.uniform(78, (1, 3, 64, 64))).to(DEV)
    with torch.no_grad():
        enc = net.encode(img)          # MIOpen convolutions are not run-to-run deterministic (1e-7):
    net.encode = lambda *a, **k: enc   # freeze the per-image stage so only the query path is compared
    try:
        vol, occ, vox_feat = ex.predict_grid(img)
        assert vol.shape == (24, 24, 24) and torch.isfinite(vol).all()
        grid = torch.tensor(utils.create_grid_points_from_bounds(-0.5, 0.5, 24)).float().unsqueeze(0).to(DEV)
        with torch.no_grad():
            one = net.query_sdf(grid, enc[0], enc[1], enc[2])
        assert torch.equal(one.view(24, 24, 24) / cfg2.sdf_scale, vol)
    finally:
        del net.encode
    net.cpu()


def test_train_entry_point_list_one_step_on_gpu(tmp_path):
    from list_amd import train as T
    cfg = arguments.default_config(model="network.models.LIST", dataset="datasets.Datasets.SyntheticIM2SDF",
                                   img_res=64, vox_res=32, sample_point_density=512, train_batch_size=2,
                                   synthetic_len=2, max_steps=1, epochs=1, output_dir=str(tmp_path) + "/",
                                   exp_name="t", load_pretrain=False)
    utils.ensure_dir(cfg.checkpoint_dir)
    loss = T.train(cfg)
    assert np.isfinite(loss)
    ck = torch.load(cfg.checkpoint_dir + "best_model_train.pt.tar", map_location="cpu")
    assert "sdf_decoder.fc.fc_0.weight" in ck["state_dict"]
