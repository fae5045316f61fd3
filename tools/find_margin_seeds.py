"""Search synth seeds whose fp32 forward keeps every ReLU pre-activation away from zero (used to pick the
gradient parity cases of oracle/cases.py).  CPU only.  usage: python tools/find_margin_seeds.py"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cases, torch_ops as TO     # noqa: E402


def min_preactivation(c):
    q, im, vx, T, W = TO.to_torch(c)
    with torch.no_grad():
        pts = q[:, :, [2, 1, 0]] * 2
        h = torch.cat((TO.stencil_voxel_features(pts, vx), TO.pooled_image_features(im, pts, T),
                       pts.transpose(1, 2)), dim=1).double()
        out = []
        for l in ("fc_0", "fc_1", "fc_2"):
            z = F.conv1d(h, W[l + ".weight"].double(), W[l + ".bias"].double())
            out.append(float(z.abs().min()))
            h = F.relu(z)
    return out


if __name__ == "__main__":
    for label, make, start in (("gtiny", lambda s: cases._case(s, 2, 129, 32, 16), 1101),
                               ("gsmall", lambda s: cases._case(s, 3, 67, 64, 32), 2202),
                               ("gedge", lambda s: cases._edge_case(s), 404)):
        for seed in range(start, start + 400):
            m = min_preactivation(make(seed))
            if min(m) > 4e-6:
                print(label, "seed", seed, "min |z| per layer", m, flush=True)
                break
