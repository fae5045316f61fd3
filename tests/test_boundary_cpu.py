"""CPU: the nn.Module / entry-point mirror of the reference API (names, state-dict keys, host logic,
config #1 plumbing = CoarseNet on CPU) against goldens produced by the reference's own models."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import fill, synth
from list_amd import arguments, utils


@pytest.fixture(scope="module")
def cfg():
    return arguments.default_config(vox_res=32, train_batch_size=2, cuda=False)


def test_get_class_resolves_reference_names():
    assert utils.get_class("network.models.LIST").__name__ == "LIST"
    assert utils.get_class("network.executors.LIST").__name__ == "LIST"
    assert utils.get_class("network.models.CoarseNet").__name__ == "CoarseNet"
    assert utils.get_class("datasets.Datasets.IM2SDF").__name__ == "IM2SDF"
    with pytest.raises(ImportError):
        utils.get_class("network.models.Nope")


def test_state_dict_names_and_shapes_match_reference(cfg, golden_dir):
    ref = json.load(open(os.path.join(golden_dir, "state_dict_keys.json")))
    for name in ("LIST", "CoarseNet"):
        mine = {k: list(v.shape) for k, v in utils.get_class(f"network.models.{name}")(cfg).state_dict().items()}
        assert mine == ref[name], set(mine) ^ set(ref[name])


def test_coarsenet_cpu_plumbing_matches_reference(cfg, golden_dir):
    """BASELINE config #1: CoarseNet, B=2, 4096 coarse points, 128^2 images, PyTorch CPU."""
    g = np.load(os.path.join(golden_dir, "models.npz"))
    net = fill.fill_state(utils.get_class("network.models.CoarseNet")(cfg), seed=1).eval()
    with torch.no_grad():
        pc = net(torch.from_numpy(synth.uniform(77, (2, 3, 128, 128))))
    assert pc.shape == (2, 4096, 3)
    np.testing.assert_allclose(pc.numpy(), g["coarse_pc"], rtol=0, atol=2e-6)


def test_list_encode_and_device_occupancy_match_reference(cfg, golden_dir):
    """Per-image stage on CPU: encoders, camera MLP and the on-device occupancy rounding (which
    replaces the reference's host KD-tree) reproduce the reference's first voxel map."""
    g = np.load(os.path.join(golden_dir, "models.npz"))
    net = fill.fill_state(utils.get_class("network.models.LIST")(cfg), seed=2).eval()
    with torch.no_grad():
        feat_l2, vox_feat, tm, pc, occ = net.encode(torch.from_numpy(synth.uniform(78, (2, 3, 64, 64))))
    assert [tuple(f.shape[1:]) for f in vox_feat] == [(1, 32, 32, 32), (16, 32, 32, 32), (32, 16, 16, 16),
                                                     (64, 8, 8, 8), (128, 4, 4, 4), (128, 2, 2, 2)]
    assert tm.shape == (2, 4, 3) and len(feat_l2) == 5
    np.testing.assert_allclose(vox_feat[0].numpy()[:, :, ::4, ::4, ::4], g["list_vox0"], rtol=0, atol=1e-5)


def test_hot_path_has_no_cpu_fallback(cfg):
    net = utils.get_class("network.models.LIST")(cfg).eval()
    with torch.no_grad(), pytest.raises(RuntimeError, match="no CPU/PyTorch fallback"):
        net(torch.rand(2, 3, 64, 64), torch.rand(2, 10, 3) - 0.5)
    from list_amd.network import modules as M
    with pytest.raises(RuntimeError, match="no CPU/PyTorch fallback"):
        M.PerceptualPooling()([torch.rand(1, 4, 8, 8)] * 5, torch.rand(1, 3, 3), torch.rand(1, 4, 3))
    assert "oracle" not in open(M.__file__).read()
    # gradients do not change that: every differentiable form goes through the C ABI (no torch-op re-evaluation)
    from list_amd.network import hotpath
    src = open(hotpath.__file__).read()
    assert "grid_sample" not in src and "conv1d" not in src and "oracle" not in src
    with pytest.raises(RuntimeError, match="no CPU/PyTorch fallback"):
        M.PerceptualPooling()([torch.rand(1, 4, 8, 8, requires_grad=True)] * 5, torch.rand(1, 3, 3),
                              torch.rand(1, 4, 3, requires_grad=True))


def test_long_queries_are_cut_at_the_backward_limit():
    from list_amd.network import hotpath
    assert hotpath._point_chunks(8, 20000) == [(0, 20000)]
    pieces = hotpath._point_chunks(1, 16777216)                   # a 256^3 grid of one image
    assert pieces[0] == (0, 262144) and pieces[-1][1] == 16777216 and len(pieces) == 64
    assert all(b - a <= 262144 for a, b in pieces) and all(p[1] == q[0] for p, q in zip(pieces, pieces[1:]))
    pieces = hotpath._point_chunks(3, 100000)                     # per-image share of the limit
    assert all(3 * (b - a) <= 262144 for a, b in pieces) and pieces[-1][1] == 100000
    with pytest.raises(RuntimeError):
        hotpath._point_chunks(300000, 1)


def test_sdf_loss_and_grid_match_reference(golden_dir):
    a = np.load(os.path.join(golden_dir, "aux.npz"))
    from list_amd.network.losses import SDFLoss
    out = SDFLoss(2.0)(torch.from_numpy(a["loss_outputs"]), torch.from_numpy(a["loss_targets"]))
    for k, v in out.items():
        np.testing.assert_allclose(v.numpy(), a["loss_" + k], rtol=2e-6)
    np.testing.assert_array_equal(utils.create_grid_points_from_bounds(-0.5, 0.5, 8), a["grid8"])
    for res in (8, 31, 128):
        host = torch.tensor(utils.create_grid_points_from_bounds(-0.5, 0.5, res)).float()
        b, e = res ** 3 // 3, res ** 3 // 3 + 4097
        dev = utils.grid_points_on_device(-0.5, 0.5, res, "cpu", b, min(e, res ** 3))
        assert torch.equal(dev, host[b:e])
    from list_amd.network import hotpath
    np.testing.assert_array_equal(hotpath.stencil_offsets("cpu").numpy(), a["displacements"])


def test_checkpoint_format_roundtrip(cfg, tmp_path):
    from list_amd.train import _Module
    net = _Module(utils.get_class("network.models.CoarseNet")(cfg))
    opt = torch.optim.Adam(net.parameters(), lr=1e-4)
    fn = str(tmp_path / "best_model_train.pt.tar")
    utils.save_checkpoint(4, net, opt, 0.25, fn)
    ck = torch.load(fn, map_location="cpu")
    assert set(ck) == {"epoch", "state_dict", "optimizer", "bestloss"} and ck["epoch"] == 5
    assert not any(k.startswith("module.") for k in ck["state_dict"])
    epoch, _, _, best = utils.load_checkpoint(fn, net, opt)
    assert (epoch, best) == (5, 0.25)


def test_synthetic_datasets_have_reference_keys(cfg):
    cfg2 = arguments.default_config(vox_res=16, img_res=32, sample_point_density=200, cuda=False)
    item = utils.get_class("datasets.Datasets.SyntheticIM2SDF")(cfg2, "train")[0]
    assert set(item) == {"rgb_image", "points", "values", "occ"}
    assert item["rgb_image"].shape == (3, 32, 32) and item["points"].shape == (200, 3)
    assert item["values"].shape == (200,) and item["occ"].shape == (1, 16, 16, 16)
    assert float(item["points"].abs().max()) <= 0.5
    pf = utils.get_class("datasets.Datasets.SyntheticIM2PointFarthest")(cfg2, "train")[1]
    assert pf["pc"].shape == (5000, 3)


def _write_shape(root, cat, shape, rng, n_views=2, img=24, with_occ=None):
    """One shape in the reference's on-disk layout (datasets/Datasets.py:176-252), .npz standing in for .h5."""
    from PIL import Image
    rgb_dir = root / "images" / cat / shape / "easy"
    h5_dir = root / "sampled_points" / cat / shape
    rgb_dir.mkdir(parents=True), h5_dir.mkdir(parents=True)
    pix = []
    for v in range(n_views):
        a = rng.randint(0, 256, (img, img, 4)).astype(np.uint8)
        Image.fromarray(a, "RGBA").save(rgb_dir / f"{v:02d}.png")
        pix.append(a)
    sig = {f"query_points_sigma_{s}": rng.randn(50 + 10 * i, 4).astype(np.float32) * 0.1
           for i, s in enumerate([0.003, 0.01, 0.07])}
    np.savez(h5_dir / "sampled_points.npz", **sig)
    pc = (rng.rand(5000, 3).astype(np.float32) - 0.5) * 0.8
    np.savez(h5_dir / "farthest_pointclouds.npz", points_5000=pc)
    if with_occ is not None:
        np.savez(h5_dir / "occupancies.npz", **with_occ)
    return pix, sig, pc


def test_file_datasets_follow_the_reference_layout(tmp_path):
    """the on-disk formats either side of the path (Datasets.py:140-304 and :56-137)."""
    rng = np.random.RandomState(7)
    split = tmp_path / "split"
    split.mkdir()
    cat = "03001627"
    (split / f"{cat}_train.lst").write_text("aaa\nbbb\nmissing\n")
    pix, sig, pc = _write_shape(tmp_path, cat, "aaa", rng)
    cached = (rng.rand(16 ** 3) < 0.1).astype(np.uint8)
    _write_shape(tmp_path, cat, "bbb", rng, with_occ={"res_16_points_5000": cached})
    cfg = arguments.default_config(vox_res=16, sample_point_density=200, cuda=False, viewnum=2, catlist=[cat],
                                   split_dir=str(split) + "/", image_dir=str(tmp_path / "images") + "/",
                                   h5_dir=str(tmp_path / "sampled_points") + "/", coarse_point_density=5000)
    ds = utils.get_class("datasets.Datasets.IM2SDF")(cfg, "train")
    assert type(ds).__name__ == "FileIM2SDF" and len(ds) == 2           # 'missing' has no files and is skipped
    counts = np.rint(np.asarray(cfg.sample_distribution) * 200).astype(int)
    item = ds[0]
    assert set(item) == {"rgb_image", "points", "values", "occ"}
    assert item["points"].shape == (counts.sum(), 3) and item["values"].shape == (counts.sum(),)
    assert item["occ"].shape == (1, 16, 16, 16) and item["rgb_image"].shape == (3, 24, 24)
    # the sampler is RandomState(333) drawing rint(distribution * density) rows per sigma, in sigma order
    ref_rng, rows = np.random.RandomState(333), []
    for s, num in zip(cfg.sigmas, counts):
        q = sig[f"query_points_sigma_{s}"]
        rows.append(q[ref_rng.randint(0, q.shape[0], num)])
    rows = np.concatenate(rows)
    np.testing.assert_array_equal(item["points"].numpy(), rows[:, :3])
    np.testing.assert_array_equal(item["values"].numpy(), rows[:, 3])
    # image planes as the reference's live reader gives them (PIL RGB -> ToTensor, Datasets.py:213-214,171):
    # red first, alpha dropped, /255
    planes = [np.ascontiguousarray(p[:, :, :3].transpose(2, 0, 1)).astype(np.float32) / 255 for p in pix]
    assert any(np.array_equal(item["rgb_image"].numpy(), p) for p in planes)
    # occupancy: nearest grid cell of every coarse point, cached next to the samples under the reference's key
    from scipy.spatial import cKDTree
    grid = utils.create_grid_points_from_bounds(cfg.bb_min, cfg.bb_max, 16)
    want = np.zeros(16 ** 3, np.float32)
    want[cKDTree(grid).query(pc)[1]] = 1
    np.testing.assert_array_equal(item["occ"].numpy().ravel(), want)
    store = np.load(tmp_path / "sampled_points" / cat / "aaa" / "occupancies.npz")
    np.testing.assert_array_equal(store["res_16_points_5000"], want.astype(np.uint8))
    np.testing.assert_array_equal(ds[1]["occ"].numpy().ravel(), cached.astype(np.float32))   # cache wins
    # the same grid the model builds from the same points on the device path (models.py:102-112)
    net = utils.get_class("network.models.LIST")(cfg)
    np.testing.assert_array_equal(net.create_occ(torch.from_numpy(pc)[None]).numpy().ravel(), want)

    pf = utils.get_class("datasets.Datasets.IM2PointFarthest")(cfg, "train")
    assert type(pf).__name__ == "FileIM2PointFarthest" and len(pf) == 2
    np.testing.assert_array_equal(pf[0]["pc"].numpy(), pc)
    img, pts = pf.get_testdata(cat, "aaa", 1)
    assert img.shape == (1, 3, 24, 24) and pts.shape == (1, 5000, 3)
    np.testing.assert_array_equal(img[0].numpy(), planes[1])
    # batches collate into what the executors read
    batch = next(iter(torch.utils.data.DataLoader(ds, batch_size=2)))
    assert batch["occ"].shape == (2, 1, 16, 16, 16) and batch["points"].shape == (2, counts.sum(), 3)
    # without split lists the reference's names give the generated stand-ins
    cfg2 = arguments.default_config(vox_res=16, cuda=False, split_dir=str(tmp_path / "nowhere") + "/")
    assert type(utils.get_class("datasets.Datasets.IM2SDF")(cfg2, "train")).__name__ == "SyntheticIM2SDF"


def test_train_entry_point_coarsenet_cpu_one_step(tmp_path):
    """train.py plumbing on CPU (config #1): one optimisation step with the synthetic loader."""
    from list_amd import train as T
    cfg = arguments.default_config(model="network.models.CoarseNet",
                                   dataset="datasets.Datasets.SyntheticIM2PointFarthest", cuda=False,
                                   img_res=128, train_batch_size=2, synthetic_len=2, max_steps=1, epochs=1,
                                   output_dir=str(tmp_path) + "/", exp_name="t", load_pretrain=False)
    utils.ensure_dir(cfg.checkpoint_dir)
    loss = T.train(cfg)
    assert np.isfinite(loss)
    assert os.path.exists(cfg.checkpoint_dir + "best_model_train.pt.tar")
