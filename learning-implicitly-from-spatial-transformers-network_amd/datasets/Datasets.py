"""The reference's datasets (datasets/Datasets.py) in two forms with the same item keys, shapes and dtypes:

  * file-backed  FileIM2SDF / FileIM2PointFarthest: the reference's on-disk layout --
      {image_dir}{cat}/{shape}/easy/{cam:02d}.png, {h5_dir}{cat}/{shape}/sampled_points.h5 with
      `query_points_sigma_{s}` [M,4] = xyz + sdf, farthest_pointclouds.h5 with `points_5000` [5000,3],
      occupancies.h5 with the cached `res_{vox}_points_{coarse}` grid, split lists {split_dir}{cat}_{status}.lst;
      sampling as Datasets.py:219-252 (RandomState(333), rint(sample_distribution * sample_point_density) per sigma);
  * synthetic    SyntheticIM2SDF / SyntheticIM2PointFarthest: generated items, no files needed.

  IM2SDF           {'rgb_image' [3,S,S] in [0,1), 'points' [N,3] in [-0.5,0.5), 'values' [N], 'occ' [1,R,R,R] in {0,1}}
  IM2PointFarthest {'rgb_image', 'pc' [5000,3]}

The reference's names IM2SDF / IM2PointFarthest pick the file-backed form when the split lists exist under the
configured directories and the synthetic one otherwise.  HDF5 needs h5py (not in this image); every `*.h5` path
falls back to a `*.npz` with the same keys, which is also what the tests use.
"""
import os
import random

import numpy as np
import torch
from torch.utils import data


class _Base(data.Dataset):
    def __init__(self, config, status="train"):
        self.config, self.status = config, status
        self.datasize = int(getattr(config, "synthetic_len", 64))
        self.img_res = config.img_res
        self.seed = 333 + (0 if status == "train" else 1)

    def __len__(self):
        return self.datasize

    def _rng(self, index):
        return np.random.RandomState(self.seed * 100003 + index)

    def _image(self, rng):
        return torch.from_numpy(rng.rand(3, self.img_res, self.img_res).astype(np.float32))


class SyntheticIM2SDF(_Base):
    def __init__(self, config, status="train"):
        super().__init__(config, status)
        dist = np.asarray(config.sample_distribution, dtype=np.float64)
        self.num_points = int(np.rint(dist * config.sample_point_density).sum())
        self.vox_res = config.vox_res
        self.max_dist = config.sdf_max_dist

    def __getitem__(self, index):
        rng = self._rng(index)
        pts = (rng.rand(self.num_points, 3).astype(np.float32) - 0.5)
        centre = (rng.rand(3).astype(np.float32) - 0.5) * 0.2
        radius = np.float32(0.2 + 0.15 * rng.rand())
        sdf = np.linalg.norm(pts - centre, axis=1).astype(np.float32) - radius
        sdf = np.clip(sdf, -self.max_dist, self.max_dist)
        axis = np.linspace(-0.5, 0.5, self.vox_res, dtype=np.float32)
        gx, gy, gz = np.meshgrid(axis, axis, axis, indexing="ij")
        shell = np.abs(np.sqrt((gx - centre[0]) ** 2 + (gy - centre[1]) ** 2 + (gz - centre[2]) ** 2)
                       - radius) < (1.0 / self.vox_res)
        return {"rgb_image": self._image(rng), "points": torch.from_numpy(pts),
                "values": torch.from_numpy(sdf), "occ": torch.from_numpy(shell.astype(np.float32)[None])}

    def get_testdata(self, cat_id=None, shape_id=None, cam_id=0):
        item = self[int(cam_id) if cam_id is not None else 0]
        return {"rgb_image": item["rgb_image"].unsqueeze(0), "gt_mesh": None}


class SyntheticIM2PointFarthest(_Base):
    def __getitem__(self, index):
        rng = self._rng(index)
        d = rng.randn(5000, 3).astype(np.float32)
        pc = d / np.linalg.norm(d, axis=1, keepdims=True) * np.float32(0.3)
        return {"rgb_image": self._image(rng), "pc": torch.from_numpy(pc)}

    def get_testdata(self, cat_id=None, shape_id=None, cam_id=0):
        item = self[int(cam_id) if cam_id is not None else 0]
        return item["rgb_image"].unsqueeze(0), item["pc"].unsqueeze(0)


# ---- the reference's on-disk layout -----------------------------------------------------------------------
class _Arrays:
    """Read-only keyed arrays: an HDF5 file through h5py, or the .npz stand-in next to it."""

    def __init__(self, path):
        self._h5 = None
        npz = os.path.splitext(path)[0] + ".npz"
        try:
            import h5py
            if os.path.exists(path):
                self._h5 = h5py.File(path, "r")
        except ImportError:
            pass
        if self._h5 is None:
            if not os.path.exists(npz):
                raise FileNotFoundError(f"{path} (h5py) / {npz} (stand-in): neither can be read")
            self._npz = np.load(npz)

    def keys(self):
        return list(self._h5.keys()) if self._h5 is not None else list(self._npz.files)

    def __getitem__(self, key):
        return np.asarray(self._h5[key]) if self._h5 is not None else self._npz[key]

    def close(self):
        if self._h5 is not None:
            self._h5.close()


def _exists_h5(path):
    return os.path.exists(path) or os.path.exists(os.path.splitext(path)[0] + ".npz")


def _read_shape_ids(filename):
    """Datasets.py:293-298: one id per line, only the newline stripped (a blank line is an id that matches no
    shape directory and drops out at the existence test, as in the reference)."""
    with open(filename) as f:
        return [line.strip("\n") for line in f.readlines()]


def _load_image(fn, config, train, rng):
    """PNG -> float32 [3,H,W] in [0,1], channels as the reference's LIVE reader yields them: PIL
    `Image.open(..).convert('RGB')` followed by T.ToTensor() (Datasets.py:45,213-214,270-271 and :35,171), i.e.
    plane 0 is RED and values are uint8 / 255.  (The cv2 `read_rgba_image` at Datasets.py:300-304 is dead code:
    its call sites at :211 and :269 are commented out.)  Normalize((0,)*3, (1,)*3) is the identity (:171-173).
    The horizontal flip is applied to the array in training when configured; colour jitter needs torchvision
    and says so."""
    from PIL import Image
    a = np.asarray(Image.open(fn).convert("RGB"), dtype=np.float32) / np.float32(255.0)
    if train and getattr(config, "random_h_flip", False) and rng.random() < 0.5:
        a = a[:, ::-1]
    if train and getattr(config, "color_jitter", False):
        raise RuntimeError("--color_jitter needs torchvision (ColorJitter(0.3, saturation=0.5, hue=0.5)), not in this image")
    return torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1)))


class _FileBase(data.Dataset):
    points_file = "sampled_points.h5"
    needs_images = False

    def __init__(self, config, status="train", max_train_shapes=2000):
        self.config, self.status = config, status
        self.rng = np.random.RandomState(333)              # Datasets.py:143
        self.viewnum = config.viewnum
        self.coarse_points = config.coarse_point_density
        split_dir = getattr(config, "split_dir", "./data/DISN_split/")
        self.datalist = []
        for cat_id in config.catlist:
            shape_ids = _read_shape_ids(os.path.join(split_dir, f"{cat_id}_{status}.lst"))
            if status == "train" and len(shape_ids) > max_train_shapes:
                shape_ids = shape_ids[:max_train_shapes]
            for shape_id in shape_ids:
                rgb_dir = os.path.join(config.image_dir, cat_id, shape_id, "easy")
                h5_fn = os.path.join(config.h5_dir, cat_id, shape_id, self.points_file)
                # IM2SDF lists a shape when its sampled_points.h5 exists (Datasets.py:186); IM2PointFarthest also
                # wants the image directory (Datasets.py:80)
                if shape_id and _exists_h5(h5_fn) and (not self.needs_images or os.path.isdir(rgb_dir)):
                    self.datalist.append({"rgba_dir": rgb_dir, "h5_fn": h5_fn, "cat_id": cat_id, "shape_id": shape_id})
        self.datasize = len(self.datalist)

    def __len__(self):
        return self.datasize

    def _image(self, rgb_dir, cam_id, train):
        return _load_image(os.path.join(rgb_dir, str(cam_id).zfill(2) + ".png"), self.config, train, random)


class FileIM2SDF(_FileBase):
    """Datasets.py:140-304."""

    def __init__(self, config, status="train"):
        super().__init__(config, status, max_train_shapes=2000)
        self.vox_res = config.vox_res
        self.query_samples = np.rint(np.asarray(config.sample_distribution) * config.sample_point_density).astype(np.uint32)
        self.sigmas = config.sigmas
        self._kdtree = None

    def create_occ(self, pc):
        """Nearest grid cell of every coarse point (Datasets.py:299-304: KD-tree over the vox_res^3 grid)."""
        if self._kdtree is None:
            from .. import utils
            self._kdtree = utils.get_kdtree(self.config.bb_min, self.config.bb_max, self.vox_res)
        occ = np.zeros(self.vox_res ** 3, dtype=np.uint8)
        _, idx = self._kdtree.query(pc)
        occ[idx] = 1
        return occ

    def _occupancy(self, h5_fn, pc):
        key = f"res_{self.vox_res}_points_{self.coarse_points}"
        occ_path = os.path.join(os.path.dirname(h5_fn), "occupancies.h5")
        occ = None
        if _exists_h5(occ_path):
            f = _Arrays(occ_path)
            if key in f.keys():
                occ = f[key]
            f.close()
        if occ is None:                                   # the reference caches it in occupancies.h5 (gzip)
            occ = self.create_occ(pc)
            try:
                import h5py
                with h5py.File(occ_path, "a") as f:
                    if key not in f.keys():
                        f.create_dataset(key, data=occ, compression="gzip")
            except ImportError:
                stand_in = os.path.splitext(occ_path)[0] + ".npz"
                old = dict(np.load(stand_in)) if os.path.exists(stand_in) else {}
                old[key] = occ
                np.savez_compressed(stand_in, **old)
        return np.asarray(occ).reshape(1, self.vox_res, self.vox_res, self.vox_res)

    def __getitem__(self, index):
        d = self.datalist[index]
        cam = random.randint(0, self.viewnum - 1)
        f = _Arrays(d["h5_fn"])
        samples = []
        for i, num in enumerate(self.query_samples):
            qdf = f["query_points_sigma_" + str(self.sigmas[i])]
            idx = self.rng.randint(0, qdf.shape[0], num)
            samples.extend(qdf[idx])
        f.close()
        samples = np.asarray(samples, dtype=np.float32)
        f = _Arrays(os.path.join(os.path.dirname(d["h5_fn"]), "farthest_pointclouds.h5"))
        pc = f["points_5000"][:]
        f.close()
        return {"rgb_image": self._image(d["rgba_dir"], cam, self.status == "train").float(),
                "points": torch.from_numpy(samples[:, :3].copy()), "values": torch.from_numpy(samples[:, 3].copy()),
                "occ": torch.from_numpy(self._occupancy(d["h5_fn"], pc).astype(np.float32))}

    def get_testdata(self, cat_id, shape_id, cam_id):
        rgb_dir = os.path.join(self.config.image_dir, cat_id, shape_id, "easy")
        mesh_fn = os.path.join(self.config.mesh_dir, cat_id, shape_id, "isosurf_scaled.obj")
        return {"rgb_image": self._image(rgb_dir, cam_id, False).unsqueeze(0), "gt_mesh": mesh_fn}


class FileIM2PointFarthest(_FileBase):
    """Datasets.py:56-137."""
    points_file = "farthest_pointclouds.h5"
    needs_images = True

    def __init__(self, config, status="train"):
        super().__init__(config, status, max_train_shapes=2500)

    def __getitem__(self, index):
        d = self.datalist[index]
        cam = random.randint(0, self.viewnum - 1)
        f = _Arrays(d["h5_fn"])
        pc = f["points_5000"][:]
        f.close()
        return {"rgb_image": self._image(d["rgba_dir"], cam, self.status == "train").float(),
                "pc": torch.from_numpy(np.asarray(pc, dtype=np.float32))}

    def get_testdata(self, cat_id, shape_id, cam_id):
        rgb_dir = os.path.join(self.config.image_dir, cat_id, shape_id, "easy")
        f = _Arrays(os.path.join(self.config.h5_dir, cat_id, shape_id, "farthest_pointclouds.h5"))
        pc = f["points_5000"][:]
        f.close()
        return self._image(rgb_dir, cam_id, False).unsqueeze(0), torch.from_numpy(np.asarray(pc, dtype=np.float32)).unsqueeze(0)


def _has_splits(config, status):
    split_dir = getattr(config, "split_dir", "./data/DISN_split/")
    return all(os.path.exists(os.path.join(split_dir, f"{c}_{status}.lst")) for c in getattr(config, "catlist", []) or ["-"])


def IM2SDF(config, status="train"):
    """The reference's name: file-backed when the split lists are there, else the synthetic generator."""
    return FileIM2SDF(config, status) if _has_splits(config, status) else SyntheticIM2SDF(config, status)


def IM2PointFarthest(config, status="train"):
    return FileIM2PointFarthest(config, status) if _has_splits(config, status) else SyntheticIM2PointFarthest(config, status)
