"""The coarse voxel levels on the matrix cores (k_gather_vox_box, gather_box_kernels.hip) against the scalar shared-tap
kernel they replaced (k_gather_vox_near, LIST_GATHER_BOX=0) at the metric's map sizes: same taps, same weights, only the
order of the fp32 sums differs -- every gathered feature within one fp16 ulp of the other kernel's (5e-7 where the taps cancel), and the SDF within the
arithmetic noise of the mode; and the MFMA kernel's own results do not depend on the point order (per-sample canonical:
sorted == unsorted bit for bit)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _run_child(tmp_path, tag, env_extra):
    out = os.path.join(tmp_path, f"box_{tag}.npz")
    env = dict(os.environ)
    env.pop("LIST_GATHER_BOX", None)
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.join(HERE, "_child_box_vs_scalar.py"), out], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return np.load(out)


def test_mfma_gather_of_the_coarse_levels_against_the_scalar_kernel(tmp_path):
    box = _run_child(str(tmp_path), "mfma", {})
    ref = _run_child(str(tmp_path), "scalar", {"LIST_GATHER_BOX": "0"})
    a, b = box["coarse_features"], ref["coarse_features"]
    assert a.shape == b.shape and a.shape[1] == 7 * 256 and np.isfinite(a).all()
    assert np.abs(b).max() > 1.0                                   # (the comparison is not vacuous)
    # X holds fp16 values: one ulp = 2^-10 relative; the two kernels' fp32 sums differ by a few 2^-24 of the LARGEST
    # summand (|voxel| up to ~4), which is more than an ulp of the result where the taps cancel: 5e-7 absolute there
    ulp = np.maximum(np.abs(b) * 2.0 ** -10, 5e-7)
    worst = float((np.abs(a - b) / ulp).max())
    differing = float((a != b).mean())
    print(f"coarse-level features: {differing:.2e} of the elements differ, worst {worst:.2f} fp16 ulp")
    assert worst <= 1.0 + 1e-6, worst
    assert differing < 5e-3, differing                            # rounding ties only (the fp32 sums differ by ~1e-7)
    for tag in ("sorted", "unsorted"):
        d = float(np.abs(box[f"sdf_{tag}"] - ref[f"sdf_{tag}"]).max())
        assert d < 2e-5, (tag, d)                                   # a few flipped fp16 roundings through the MLP
    # per-sample canonical arithmetic: the point order (Morton / none) does not change a bit, in either kernel
    np.testing.assert_array_equal(box["sdf_sorted"], box["sdf_unsorted"])
    np.testing.assert_array_equal(ref["sdf_sorted"], ref["sdf_unsorted"])
