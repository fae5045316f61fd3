"""Synthetic on-disk shapes in the reference's dataset layout (TEST INFRASTRUCTURE).

One writer shared by oracle/gen_golden.py (which runs the REFERENCE's datasets.Datasets.IM2SDF /
IM2PointFarthest over the tree, through the real split lists under /root/reference/data/DISN_split) and by
tests/test_dataset_golden.py (which runs this package's readers over an identical tree): both sides see the same
bytes without committing them.  Layout (reference Datasets.py:176-196,219-252, preprocessing/preprocess.py:101-111,
farthest_pointcloud.py:30-31):

    <image_dir>/<cat>/<shape>/easy/<cam:02d>.png                       RGBA renderings
    <h5_dir>/<cat>/<shape>/sampled_points.h5      query_points_sigma_{0.003,0.01,0.07} [M,4] = xyz + sdf
    <h5_dir>/<cat>/<shape>/farthest_pointclouds.h5  points_5000 [5000,3]
    <h5_dir>/<cat>/<shape>/occupancies.h5         res_{vox}_points_{coarse} uint8 (cache, written on first use)

h5py is not in this image: arrays are stored as .npz next to the .h5 name with the SAME keys (the package's
readers take either; the generator gives the reference an h5py stand-in over the same files).
"""
import os

import numpy as np

from . import synth

CAT = "03001627"
N_SHAPES = 3                 # the first ids of the reference's own 03001627_train.lst
N_VIEWS = 2
IMG = 20
SIGMAS = (0.003, 0.01, 0.07)
ROWS = (60, 75, 90)          # rows per sigma table (different, so a mixed-up table shows)
SAMPLE_POINT_DENSITY = 200
COARSE_POINTS = 5000
VOX_RES = 16


def write_tree(root, shape_ids):
    """Writes the shapes; returns (image_dir, h5_dir) with trailing slashes like the reference's config."""
    image_dir, h5_dir = os.path.join(root, "images") + "/", os.path.join(root, "sampled_points") + "/"
    from PIL import Image
    for si, shape in enumerate(shape_ids):
        rgb = os.path.join(image_dir, CAT, shape, "easy")
        h5 = os.path.join(h5_dir, CAT, shape)
        os.makedirs(rgb, exist_ok=True)
        os.makedirs(h5, exist_ok=True)
        for v in range(N_VIEWS):
            a = (synth.uniform(7000 + 10 * si + v, (IMG, IMG, 4)) * 256).astype(np.uint8)
            Image.fromarray(a, "RGBA").save(os.path.join(rgb, f"{v:02d}.png"))
        tables = {f"query_points_sigma_{s}": synth.normalish(7100 + 10 * si + i, (ROWS[i], 4), 0.1)
                  for i, s in enumerate(SIGMAS)}
        np.savez(os.path.join(h5, "sampled_points.npz"), **tables)
        pc = synth.uniform(7200 + si, (COARSE_POINTS, 3), -0.4, 0.4)
        np.savez(os.path.join(h5, "farthest_pointclouds.npz"), points_5000=pc)
        for name in ("sampled_points.h5", "farthest_pointclouds.h5"):       # the reference tests os.path.exists(.h5)
            open(os.path.join(h5, name), "wb").close()
    return image_dir, h5_dir


def config_fields(image_dir, h5_dir):
    """The config attributes both dataset implementations read (reference Datasets.py:141-175)."""
    return dict(catlist=[CAT], viewnum=1, sampling_mode="weighted", sample_point_density=SAMPLE_POINT_DENSITY,
                coarse_point_density=COARSE_POINTS, vox_res=VOX_RES, sample_distribution=[0.5, 0.49, 0.01],
                sigmas=list(SIGMAS), bb_min=-0.5, bb_max=0.5, image_dir=image_dir, h5_dir=h5_dir,
                random_h_flip=False, color_jitter=False, normalize=True)
