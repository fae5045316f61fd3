"""Synthetic stand-ins for the reference's datasets (datasets/Datasets.py): same item keys, shapes,
dtypes and value ranges as IM2SDF (:140-304) and IM2PointFarthest (:56-137), no files needed.

  SyntheticIM2SDF           {'rgb_image' [3,S,S] in [0,1), 'points' [N,3] in [-0.5,0.5),
                             'values' [N] (signed distance to a random sphere, clamped),
                             'occ' [R,R,R] in {0,1}}
  SyntheticIM2PointFarthest {'rgb_image', 'pc' [5000,3]}
"""
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
                "values": torch.from_numpy(sdf), "occ": torch.from_numpy(shell.astype(np.float32))}

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


# the reference's names resolve to the synthetic generators when no data directory is configured
IM2SDF = SyntheticIM2SDF
IM2PointFarthest = SyntheticIM2PointFarthest
