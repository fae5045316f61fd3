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


def test_mfma_gather_of_the_coarse_levels_against_the_oracle():
    """The matrix-core gather against the ORACLE (oracle.list_oracle.vox_features = network/modules.py:256-273 in numpy),
    not against another HIP kernel: the 16^3 and 8^3 x 128-channel levels at the metric's map sizes, points on faces and
    corners included.  The oracle samples the fp16-ROUNDED maps (what the prepared levels hold) in fp32; X then holds
    that value rounded to fp16: one fp16 ulp (2^-10 relative) + 5e-7 where the taps cancel."""
    import torch
    import __graft_entry__ as ge
    ge.build()
    from list_amd import hip
    from list_amd import synthetic as synth
    from oracle import list_oracle as O            # the checker
    assert os.environ.get("LIST_GATHER_BOX", "1") != "0"
    seed, B, N = 808, 2, 6000
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
    q = synth.make_query(seed, B, N)
    q[0, :8] = np.array([[s0, s1, s2] for s0 in (-0.5, 0.5) for s1 in (-0.5, 0.5) for s2 in (-0.5, 0.5)], np.float32)
    q[1, :3] = np.array([[0.5, 0.1, -0.2], [0.0, -0.5, 0.3], [0.49, 0.49, 0.0]], np.float32)
    vox_maps = synth.make_vox_maps(seed, B, 128)
    img = hip.prep_img_maps([dev(m) for m in synth.make_img_maps(seed, B, 224)], dtype="f16")
    vox = hip.prep_vox_maps([dev(m) for m in vox_maps], dtype="f16")
    packed = hip.prep_mlp_weights({k: dev(v) for k, v in synth.make_mlp_weights(seed).items()}, vox.channels,
                                  img.channels, "fp16")
    tm = dev(synth.make_trans_mat(seed, B))
    # the library's own word on what it dispatches for these arguments: levels 4 and 5 on the matrix cores
    plan = {}
    hip.sdf_query(dev(q), tm, img, vox, packed, precision="fp16", plan=plan)
    assert plan["box_levels"] == (1 << 4) | (1 << 5), plan
    feats = hip.gather_features(dev(q), tm, img, vox, packed).cpu().numpy()
    lo = 7 * (1 + 16 + 32 + 64)
    got = feats[:, lo:lo + 7 * 256]
    coarse16 = [m.astype(np.float16).astype(np.float32) for m in vox_maps[4:6]]
    assert coarse16[0].shape[1:] == (128, 16, 16, 16) and coarse16[1].shape[1:] == (128, 8, 8, 8)
    want = O.vox_features(O.permute_scale_query(q), coarse16)          # [B, 256 * 7, N], k = c * 7 + j
    assert want.shape == got.shape and np.abs(want).max() > 1.0
    tol = np.abs(want) * 2.0 ** -10 + 5e-7
    worst = float((np.abs(got - want) / tol).max())
    print(f"box gather vs oracle: worst {worst:.3f} of (1 fp16 ulp + 5e-7)")
    assert worst <= 1.0, worst
    # the face / corner points sit on border taps (clipped coordinates, dropped taps): checked separately
    edge = np.abs(got[0, :, :8] - want[0, :, :8]) / tol[0, :, :8]
    assert float(edge.max()) <= 1.0
