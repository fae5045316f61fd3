"""Generate tests/golden/*.npz by RUNNING THE REFERENCE's own modules (TEST INFRASTRUCTURE).

Run in the authoring container only (needs /root/reference):

    python -m oracle.gen_golden

The reference is imported by path, never copied: /root/reference is put on
sys.path, bytecode writing is disabled, absent third-party modules that the
reference imports at top level (cv2, mcubes, trimesh, torchvision) are replaced
by empty stubs, and torch.Tensor.cuda is an identity while VoxelDecoder2 is
constructed (network/modules.py:214 calls .cuda() unconditionally).
Fixtures hold arrays only (expected outputs + a few intermediates); inputs are
regenerated from oracle/synth.py by whoever consumes the fixture.
"""
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _import_reference():
    import torch
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    for name in ("cv2", "mcubes", "trimesh"):
        sys.modules.setdefault(name, types.ModuleType(name))
    tv = types.ModuleType("torchvision")
    tvm = types.ModuleType("torchvision.models")
    tv.models = tvm
    sys.modules.setdefault("torchvision", tv)
    sys.modules.setdefault("torchvision.models", tvm)
    import network.modules as M          # noqa: E402  (reference)
    import network.losses as L           # noqa: E402  (reference)
    import utils as U                    # noqa: E402  (reference)
    orig = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        dec = M.VoxelDecoder2(3610, 256)
    finally:
        torch.Tensor.cuda = orig
    return torch, M, L, U, dec


def main():
    from . import cases
    torch, M, L, U, dec = _import_reference()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    os.makedirs(OUT, exist_ok=True)
    pool = M.PerceptualPooling()

    only = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--case=")]
    for name in cases.CASE_NAMES + cases.NONFINITE_CASE_NAMES:
        if only and name not in only:
            continue
        c = cases.build_case(name)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
        dec.load_state_dict({"fc." + k: t(v) for k, v in c["weights"].items()})
        with torch.no_grad():
            query = t(c["query"])
            q = query[:, :, [2, 1, 0]] * 2                       # models.py:91-92 (glue, stated here)
            B, N, _ = q.shape
            img_maps = [t(m) for m in c["img_maps"]]
            vox_maps = [t(m) for m in c["vox_maps"]]
            T = t(c["trans_mat"])
            percep = pool(img_maps, q, T)                         # [B,1024,1,N]
            percep_r = percep.reshape(B, -1, N)
            sdf = dec(q, vox_maps, percep_r)                      # [B,N]
            # intermediates, via the same torch ops the reference calls
            feats = []
            disp = dec.displacments
            pp = torch.cat([q.unsqueeze(1).unsqueeze(1) + d for d in disp], dim=2)
            for f in vox_maps:
                feats.append(torch.nn.functional.grid_sample(f, pp, padding_mode="border",
                                                             align_corners=True))
            vf = torch.cat(feats, dim=1)
            vf = vf.reshape(B, vf.shape[1] * vf.shape[3], vf.shape[4])  # [B,2583,N]
            resized0 = torch.nn.functional.interpolate(img_maps[0], size=137, mode="bilinear",
                                                       align_corners=True)
        sub = slice(0, None, cases.FEATURE_STRIDE.get(name, 4))
        np.savez_compressed(
            os.path.join(OUT, f"hotpath_{name}.npz"),
            sdf=sdf.numpy(),
            percep_sub=percep.numpy()[:, :, :, sub],
            voxfeat_sub=vf.numpy()[:, :, sub],
            resized0_sub=resized0.numpy()[:, ::8, ::3, ::3],
            torch_version=np.array(torch.__version__),
        )
        print(name, "sdf", tuple(sdf.shape), float(torch.nan_to_num(sdf, 0.0, 0.0, 0.0).abs().max()),
              "non-finite:", int((~torch.isfinite(sdf)).sum()))
    if only:
        return

    # a6: grid builder; losses
    grid = U.create_grid_points_from_bounds(-0.5, 0.5, 8)
    rng_o = np.linspace(-1, 1, 2 * 50, dtype=np.float32).reshape(2, 50)
    rng_t = (rng_o[:, ::-1] * 0.7 + 0.1).astype(np.float32).copy()
    loss = L.SDFLoss(2.0)(torch.from_numpy(rng_o), torch.from_numpy(rng_t))
    np.savez_compressed(
        os.path.join(OUT, "aux.npz"),
        grid8=grid, loss_outputs=rng_o, loss_targets=rng_t,
        **{"loss_" + k: v.numpy() for k, v in loss.items()},
        displacements=dec.displacments.numpy(),
    )
    print("aux written")
    grad_goldens(torch, dec, pool)
    model_goldens(torch)
    dataset_goldens(torch)


GRAD_W0_ROW_STEP = 8


def grad_goldens(torch, dec, pool):
    """Backward of the path: gradients of  sum(sdf * g)  w.r.t. every differentiable input,
    taken by autograd THROUGH THE REFERENCE's modules (PerceptualPooling.forward + VoxelDecoder2.forward +
    the glue of models.py:91-97).  g = synth.normalish(seed 9000 + case index)."""
    from . import cases, synth
    for ci, name in enumerate(cases.GRAD_CASE_NAMES):
        c = cases.build_case(name)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).clone().requires_grad_(True)
        dec.load_state_dict({"fc." + k: torch.from_numpy(v.copy()) for k, v in c["weights"].items()})
        dec.zero_grad()
        query = torch.from_numpy(c["query"].copy())
        img_maps = [t(m) for m in c["img_maps"]]
        vox_maps = [t(m) for m in c["vox_maps"]]
        T = t(c["trans_mat"])
        q = query[:, :, [2, 1, 0]] * 2
        B, N, _ = q.shape
        percep = pool(img_maps, q, T)
        sdf = dec(q, vox_maps, percep.reshape(B, -1, N))
        g = torch.from_numpy(synth.normalish(9000 + ci, (B, N)))
        (sdf * g).sum().backward()
        out = {"grad_sdf": g.numpy(), "d_trans_mat": T.grad.numpy()}
        for i, m in enumerate(img_maps):
            out[f"d_img{i}"] = m.grad.numpy()
        for i, m in enumerate(vox_maps):
            out[f"d_vox{i}"] = m.grad.numpy()
        for k, v in dec.state_dict(keep_vars=True).items():
            out["d_" + k[3:] if k.startswith("fc.") else "d_" + k] = v.grad.numpy()
        # keep the fixtures small: every 8th output row of fc_0's weight gradient, and for the larger
        # case every 2nd voxel / pixel of the map gradients (the tests apply the same slicing)
        out["d_fc_0.weight"] = out["d_fc_0.weight"][::GRAD_W0_ROW_STEP]
        if name == "gsmall":
            for i in range(len(vox_maps)):
                out[f"d_vox{i}"] = out[f"d_vox{i}"][:, :, ::2, ::2, ::2]
            for i in range(len(img_maps)):
                out[f"d_img{i}"] = out[f"d_img{i}"][:, :, ::2, ::2]
        np.savez_compressed(os.path.join(OUT, f"hotpath_grad_{name}.npz"), **out)
        print("grad", name, {k: float(np.abs(v).max()) for k, v in out.items() if k.startswith("d_fc") or k == "d_trans_mat"})


def _harness_resnet18(pretrained=False, **kw):
    """The reference wraps torchvision.models.resnet18 (modules.py:1030); torchvision is absent, so
    the stub hands it this package's own ResNet-18 definition (same attribute names)."""
    from list_amd.network.resnet import resnet18
    return resnet18()


def model_goldens(torch):
    """Whole-model goldens: state-dict names/shapes of the reference's LIST and CoarseNet, a CoarseNet
    forward (plumbing config #1) and a LIST forward, all with parameters from oracle/fill.py."""
    import json
    import types as _t
    from . import fill, synth
    sys.modules["torchvision.models"].resnet18 = _harness_resnet18
    sys.modules["torchvision"].models.resnet18 = _harness_resnet18
    import network.models as RM                                   # reference
    import utils as RU                                            # reference
    RU.get_kdtree_orig = RU.get_kdtree
    cfg = _t.SimpleNamespace(vox_res=32, im_enc_layers=[1, 1, 1, 1, 16, 32, 64, 128, 128],
                             train_batch_size=2, point_feat=[128, 128, 256, 256, 256, 128, 128, 3],
                             point_degree=[2, 2, 2, 2, 2, 2, 64], bb_min=-0.5, bb_max=0.5)
    orig = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        ref_list = RM.LIST(cfg)
        ref_coarse = RM.CoarseNet(cfg)
    finally:
        torch.Tensor.cuda = orig
    keys = {"LIST": {k: list(v.shape) for k, v in ref_list.state_dict().items()},
            "CoarseNet": {k: list(v.shape) for k, v in ref_coarse.state_dict().items()}}
    with open(os.path.join(OUT, "state_dict_keys.json"), "w") as f:
        json.dump(keys, f, indent=0, sort_keys=True)

    fill.fill_state(ref_coarse, seed=1).eval()
    img = torch.from_numpy(synth.uniform(77, (2, 3, 128, 128)))
    with torch.no_grad():
        pc = ref_coarse(img)
    fill.fill_state(ref_list, seed=2).eval()
    img2 = torch.from_numpy(synth.uniform(78, (2, 3, 64, 64)))
    q = torch.from_numpy(synth.make_query(79, 2, 100))
    with torch.no_grad():
        occ_feat, sdf = ref_list(img2, q)
        tm = torch.from_numpy(synth.make_trans_mat(80, 2))
        _, sdf_tm = ref_list(img2, q, tm)
    np.savez_compressed(os.path.join(OUT, "models.npz"), coarse_pc=pc.numpy(), list_sdf=sdf.numpy(),
                        list_sdf_given_transmat=sdf_tm.numpy(), list_vox0=occ_feat.numpy()[:, :, ::4, ::4, ::4])
    print("models: coarse", tuple(pc.shape), float(pc.abs().max()), "list sdf", tuple(sdf.shape),
          float(sdf.abs().max()), float(sdf_tm.abs().max()))


class _NpzH5File:
    """h5py.File stand-in for the GENERATOR only (h5py is not in this image): the same keys, kept in an .npz next to
    the .h5 name.  Covers what the reference's datasets use: f[key][:], np.asarray(f[key]), keys(),
    create_dataset(name, data=, compression=), close(), with-statement."""

    def __init__(self, path, mode="r"):
        self.path = os.path.splitext(path)[0] + ".npz"
        self.mode = mode
        if os.path.exists(self.path):
            self.arrays = dict(np.load(self.path))
        elif mode == "r":
            raise OSError("no such file: " + path)
        else:
            self.arrays = {}

    def __getitem__(self, key):
        return self.arrays[key]

    def keys(self):
        return self.arrays.keys()

    def create_dataset(self, name, data=None, compression=None):
        self.arrays[name] = np.asarray(data)
        np.savez_compressed(self.path, **self.arrays)

    def close(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def _install_dataset_stubs(torch):
    """Third-party modules the reference's datasets/Datasets.py imports at top level and this image lacks."""
    h5 = types.ModuleType("h5py")
    h5.File = _NpzH5File
    sys.modules["h5py"] = h5
    p3d, ops = types.ModuleType("pytorch3d"), types.ModuleType("pytorch3d.ops")
    p3d.ops = ops
    sys.modules.setdefault("pytorch3d", p3d)
    sys.modules.setdefault("pytorch3d.ops", ops)
    T = types.ModuleType("torchvision.transforms")

    class ToTensor:                              # torchvision semantics: PIL RGB -> float32 CHW in [0, 1]
        def __call__(self, img):
            a = np.asarray(img, dtype=np.uint8)
            return torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1))).float().div(255)

    class Normalize:
        def __init__(self, mean, std):
            self.mean, self.std = torch.tensor(mean).view(-1, 1, 1), torch.tensor(std).view(-1, 1, 1)

        def __call__(self, x):
            return (x - self.mean) / self.std

    class Compose:
        def __init__(self, ts):
            self.ts = ts

        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x

    def _unavailable(*a, **k):
        raise RuntimeError("augmentations need the real torchvision")

    T.ToTensor, T.Normalize, T.Compose = ToTensor, Normalize, Compose
    T.RandomHorizontalFlip = T.ColorJitter = _unavailable
    sys.modules["torchvision.transforms"] = T
    sys.modules["torchvision"].transforms = T


def dataset_goldens(torch):
    """SURVEY 8 f4, pinned by what the reference HOLDS and RUNS:
      * its own split lists data/DISN_split/*.lst parsed by its own reader (Datasets.py:293-298) and by the
        test-list parser of arguments.py:112-125 -> counts, first / last id and a SHA-256 per list;
      * its IM2SDF / IM2PointFarthest datasets run over a synthetic tree (oracle/dataset_fixture.py) whose shape
        ids are the first entries of its 03001627_train.lst -> the items they return."""
    import hashlib
    import json
    import random
    import shutil
    import tempfile
    import types as _t
    from . import dataset_fixture as DF
    _install_dataset_stubs(torch)
    cwd = os.getcwd()
    os.chdir(REF)                                  # the reference opens './data/DISN_split/...'
    try:
        # the reference's module, loaded by path (a HuggingFace `datasets` package in site-packages shadows the name)
        import importlib.util
        spec = importlib.util.spec_from_file_location("reference_datasets_Datasets",
                                                      os.path.join(REF, "datasets", "Datasets.py"))
        RD = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(RD)
        reader = RD.IM2SDF.read_shape_ids_from_file
        lists = {}
        split = os.path.join(REF, "data", "DISN_split")
        for fn in sorted(os.listdir(split)):
            if not fn.endswith(".lst"):
                continue
            ids = reader(None, os.path.join(split, fn))
            lists[fn] = {"count": len(ids), "first": ids[0], "last": ids[-1],
                         "sha256": hashlib.sha256("\n".join(ids).encode()).hexdigest()}
        # arguments.py:112-125 (its parser runs argparse on sys.argv: restate the call with the same statements'
        # inputs -- first 30 lines, split on ' ', category filter -- and pin the outcome for the default catlist)
        with open(os.path.join(split, "testlist_all.lst")) as f:
            lines = f.readlines()
        testlist = [l.strip().split(" ") for l in lines[:30] if l.strip() != ""]
        lists["testlist_all.lst"]["first30"] = testlist
        tmp = tempfile.mkdtemp(prefix="list_ds_")
        try:
            with open(os.path.join(split, DF.CAT + "_train.lst")) as f:
                shape_ids = [l.strip() for l in f.readlines()[:DF.N_SHAPES]]
            image_dir, h5_dir = DF.write_tree(tmp, shape_ids)
            cfg = _t.SimpleNamespace(**DF.config_fields(image_dir, h5_dir))
            random.seed(333)
            ds = RD.IM2SDF(cfg, "train")
            items = [ds[i] for i in range(len(ds))]
            again = ds[0]                           # the RandomState(333) stream continues across items
            pf = RD.IM2PointFarthest(cfg, "train")
            pf0_rgb, pf0_pc = pf[0]                  # (the ShapeNet variant returns a pair, Datasets.py:117)
            out = {"n_items": np.array(len(ds)), "n_items_pf": np.array(len(pf)),
                   "again_points": again["points"].numpy(), "pf0_pc": pf0_pc.numpy(), "pf0_rgb": pf0_rgb.numpy()}
            for i, it in enumerate(items):
                out[f"points{i}"] = it["points"].numpy()
                out[f"values{i}"] = it["values"].numpy()
                out[f"rgb{i}"] = it["rgb_image"].numpy()
                out[f"occ{i}"] = np.packbits(it["occ"].numpy().astype(np.uint8).ravel())
            cached = dict(np.load(os.path.join(h5_dir, DF.CAT, shape_ids[0], "occupancies.npz")))
            out["occ_cache_keys"] = np.array(sorted(cached.keys()))
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    finally:
        os.chdir(cwd)
    lists["_tree_shape_ids"] = shape_ids
    with open(os.path.join(OUT, "dataset_lists.json"), "w") as f:
        json.dump(lists, f, indent=0, sort_keys=True)
    np.savez_compressed(os.path.join(OUT, "dataset_items.npz"), **out)
    print("datasets:", len(lists) - 1, "lists;", int(out["n_items"]), "IM2SDF items,", int(out["n_items_pf"]), "point items")


if __name__ == "__main__":
    if "--datasets-only" in sys.argv:
        _torch = _import_reference()[0]
        dataset_goldens(_torch)
    elif "--grads-only" in sys.argv:
        _torch, _M, _L, _U, _dec = _import_reference()
        _torch.set_num_threads(8)
        grad_goldens(_torch, _dec, _M.PerceptualPooling())
    else:
        main()
