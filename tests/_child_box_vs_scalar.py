"""Child process of tests/test_box_gather_gpu.py: the gathered features and the SDF of the metric-size case with the
coarse voxel levels on the matrix cores (default) or on the scalar shared-tap kernel (LIST_GATHER_BOX=0, read when the
library takes its first coarse level: hence a fresh process each), written to an .npz."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from list_amd import synthetic as synth          # noqa: E402


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


def main(out_path):
    import __graft_entry__ as ge
    ge.build()
    from list_amd import hip
    seed, B, N = 808, 2, 6000                    # real map sizes (128^3 pyramid, 224^2 images), 12 000 points
    c = {"query": synth.make_query(seed, B, N), "img_maps": synth.make_img_maps(seed, B, 224),
         "vox_maps": synth.make_vox_maps(seed, B, 128), "weights": synth.make_mlp_weights(seed),
         "trans_mat": synth.make_trans_mat(seed, B)}
    # a few points on the faces and corners of the box: border taps of the coarse levels
    q = c["query"].copy()
    q[0, :8] = np.array([[s0, s1, s2] for s0 in (-0.5, 0.5) for s1 in (-0.5, 0.5) for s2 in (-0.5, 0.5)], np.float32)
    q[1, :3] = np.array([[0.5, 0.1, -0.2], [0.0, -0.5, 0.3], [0.49, 0.49, 0.0]], np.float32)
    res = {}
    img = hip.prep_img_maps([dev(m) for m in c["img_maps"]], dtype="f16")
    vox = hip.prep_vox_maps([dev(m) for m in c["vox_maps"]], dtype="f16")
    packed = hip.prep_mlp_weights({k: dev(v) for k, v in c["weights"].items()}, vox.channels, img.channels, "fp16")
    for sort in (True, False):
        tag = "sorted" if sort else "unsorted"
        res[f"sdf_{tag}"] = hip.sdf_query(dev(q), dev(c["trans_mat"]), img, vox, packed, precision="fp16",
                                          sort_points=sort).cpu().numpy()
    feats = hip.gather_features(dev(q), dev(c["trans_mat"]), img, vox, packed).cpu().numpy()   # [B, F, N] (fp16 X: packed.fp16)
    lo = 7 * (1 + 16 + 32 + 64)                   # reference order: (cbase + c) * 7 + j -> the two 128-channel levels
    res["coarse_features"] = feats[:, lo:lo + 7 * 256]
    np.savez(out_path, **res)


if __name__ == "__main__":
    main(sys.argv[1])
