"""GPU debug run of the backward: prints the normalised error of every gradient against the golden
fixtures (reference autograd) for each case and precision.  Not a test; see tests/test_hip_backward.py."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import cases, synth                      # noqa: E402
from list_amd import hip                             # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_oracle_golden import slice_like_golden     # noqa: E402


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


def run(name, precision, sort_points=True):
    c = cases.build_case(name)
    g = np.load(os.path.join(ROOT, "tests", "golden", f"hotpath_grad_{name}.npz"))
    md = hip.map_dtype_for(precision)
    img_in = [dev(m) for m in c["img_maps"]]
    img = hip.prep_img_maps(img_in, dtype=md)
    vox = hip.prep_vox_maps([dev(m) for m in c["vox_maps"]], dtype=md)
    params = {k: dev(v) for k, v in c["weights"].items()}
    packed = hip.prep_mlp_weights(params, vox.channels, img.channels, precision)
    packed_b = hip.prep_mlp_weights_bwd(params, vox.channels, img.channels, precision)
    sdf, ctx = hip.sdf_query(dev(c["query"]), dev(c["trans_mat"]), img, vox, packed, precision=precision,
                             save_for_backward=True, sort_points=sort_points)
    out = hip.sdf_query_backward(ctx, dev(g["grad_sdf"]), packed_b)
    torch.cuda.synchronize()
    lv = hip.img_map_grad_to_levels(out["img_map"], img_in)
    torch.cuda.synchronize()
    got = {"d_trans_mat": out["trans_mat"]}
    got.update({"d_" + k: v for k, v in out["mlp"].items()})
    got.update({f"d_vox{i}": v.permute(0, 4, 1, 2, 3) for i, v in enumerate(out["vox"])})
    got.update({f"d_img{i}": v for i, v in enumerate(lv)})
    worst = 0.0
    for k in sorted(g.files):
        if not k.startswith("d_"):
            continue
        a = slice_like_golden(name, k, got[k].cpu().numpy())
        ref = g[k]
        err = float(np.abs(a - ref).max()) / max(float(np.abs(ref).max()), 1e-12)
        l2 = float(np.linalg.norm((a - ref).ravel())) / max(float(np.linalg.norm(ref.ravel())), 1e-30)
        worst = max(worst, err)
        print(f"  {name:6s} {precision:7s} sort={int(sort_points)} {k:16s} rel-max-err {err:.3e} rel-l2 {l2:.3e}"
              f"  (|ref|max {np.abs(ref).max():.3e}, nan {int(np.isnan(a).sum())})", flush=True)
    return worst


if __name__ == "__main__":
    names = sys.argv[1].split(",") if len(sys.argv) > 1 else ["gtiny", "gsmall", "gedge"]
    precs = sys.argv[2].split(",") if len(sys.argv) > 2 else ["bf16x3", "fp16", "bf16"]
    for n in names:
        for p in precs:
            print(n, p, "worst", run(n, p))
    print("nosort", run("gtiny", "bf16x3", sort_points=False))
