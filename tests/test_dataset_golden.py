"""CPU: the on-disk formats either side of the path against what the REFERENCE holds and runs.

tests/golden/dataset_lists.json and dataset_items.npz come from oracle/gen_golden.py --datasets-only, which parsed
the reference's own split lists (data/DISN_split/*.lst) with the reference's reader and ran the reference's
datasets.Datasets.IM2SDF / IM2PointFarthest over the synthetic tree of oracle/dataset_fixture.py.  Here this
package's readers see the same lists and an identical tree."""
import hashlib
import json
import os
import random

import numpy as np
import pytest

from list_amd import arguments, utils
from list_amd.datasets import Datasets as D
from oracle import dataset_fixture as DF

SPLIT = "/root/reference/data/DISN_split"


@pytest.fixture(scope="module")
def lists(golden_dir):
    return json.load(open(os.path.join(golden_dir, "dataset_lists.json")))


@pytest.mark.skipif(not os.path.isdir(SPLIT), reason="the reference's split lists live in the authoring container only")
def test_split_lists_parse_like_the_reference_reader(lists):
    """Datasets.py:293-298 over every list the reference ships: same ids, same order."""
    n = 0
    for fn, want in lists.items():
        if fn.startswith("_"):
            continue
        ids = D._read_shape_ids(os.path.join(SPLIT, fn))
        assert len(ids) == want["count"] and ids[0] == want["first"] and ids[-1] == want["last"], fn
        assert hashlib.sha256("\n".join(ids).encode()).hexdigest() == want["sha256"], fn
        n += 1
    assert n == 27
    # arguments.py:112-125: the first 30 lines of testlist_all.lst, category-filtered
    cfg = arguments.default_config(testlist_file=os.path.join(SPLIT, "testlist_all.lst"))
    got = [[t["cat_id"], t["shape_id"], t["cam_id"]] for t in cfg.testlist]
    want = [t for t in lists["testlist_all.lst"]["first30"] if t[0] in cfg.catlist]
    assert got == want and len(got) > 0


def test_file_datasets_return_what_the_reference_datasets_return(tmp_path, golden_dir, lists):
    """Per-sigma sampling order under RandomState(333), key names, row counts rint(distribution * density),
    RGB plane order and /255, the occupancy grid and its cache key (Datasets.py:140-304, 56-137)."""
    g = np.load(os.path.join(golden_dir, "dataset_items.npz"))
    shape_ids = lists["_tree_shape_ids"]
    image_dir, h5_dir = DF.write_tree(str(tmp_path), shape_ids)
    split = tmp_path / "split"
    split.mkdir()
    # the head of the reference's own 03001627_train.lst, plus an id without files (skipped like there)
    (split / f"{DF.CAT}_train.lst").write_text("\n".join(shape_ids + ["0000_no_such_shape"]) + "\n")
    cfg = arguments.default_config(cuda=False, split_dir=str(split) + "/", **DF.config_fields(image_dir, h5_dir))
    random.seed(333)
    ds = utils.get_class("datasets.Datasets.IM2SDF")(cfg, "train")
    assert type(ds).__name__ == "FileIM2SDF" and len(ds) == int(g["n_items"]) == DF.N_SHAPES
    items = [ds[i] for i in range(len(ds))]
    for i, it in enumerate(items):
        np.testing.assert_array_equal(it["points"].numpy(), g[f"points{i}"])
        np.testing.assert_array_equal(it["values"].numpy(), g[f"values{i}"])
        np.testing.assert_array_equal(it["rgb_image"].numpy(), g[f"rgb{i}"])
        occ = np.unpackbits(g[f"occ{i}"])[:DF.VOX_RES ** 3].reshape(1, DF.VOX_RES, DF.VOX_RES, DF.VOX_RES)
        np.testing.assert_array_equal(it["occ"].numpy(), occ.astype(np.float32))
    np.testing.assert_array_equal(ds[0]["points"].numpy(), g["again_points"])        # the sampler's stream goes on
    cache = np.load(os.path.join(h5_dir, DF.CAT, shape_ids[0], "occupancies.npz"))
    assert sorted(cache.files) == list(g["occ_cache_keys"])
    pf = utils.get_class("datasets.Datasets.IM2PointFarthest")(cfg, "train")
    assert len(pf) == int(g["n_items_pf"])
    first = pf[0]
    np.testing.assert_array_equal(first["pc"].numpy(), g["pf0_pc"])
    np.testing.assert_array_equal(first["rgb_image"].numpy(), g["pf0_rgb"])
