"""Dev check (GPU): sdf with the low-resolution encoder levels projected through fc_0 before the resize
(hip.prep_img_proj, list_prep_img_proj) against the standard path, and the time of both, per precision."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ge.build()
from list_amd import hip  # noqa: E402
from list_amd import synthetic as synth  # noqa: E402
from oracle import cases  # noqa: E402


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


def timed(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


def one(c, tag, ms=137, hi=136.0, precisions=("fp16", "bf16x3", "bf16"), time_it=False, n_kept=None):
    for prec in precisions:
        md = hip.map_dtype_for(prec)
        maps = [dev(m) for m in c["img_maps"]]
        voxm = [dev(m) for m in c["vox_maps"]]
        w = {k: dev(v) for k, v in c["weights"].items()}
        q, T = dev(c["query"]), dev(c["trans_mat"])

        def std(fused=True):
            img = hip.prep_img_maps(maps, ms, md)
            vox = hip.prep_vox_maps(voxm, md)
            packed = hip.prep_mlp_weights(w, vox.channels, img.channels, prec)
            return hip.sdf_query(q, T, img, vox, packed, precision=prec, clamp_hi=hi, fused_fc0=fused)

        plan = {}

        def prj():
            vox = hip.prep_vox_maps(voxm, md)
            packed = hip.prep_mlp_weights(w, vox.channels, sum(m.shape[1] for m in maps), prec)
            img = hip.prep_img_proj(maps, packed, ms, prec, n_kept_levels=n_kept)
            return hip.sdf_query(q, T, img, vox, packed, precision=prec, clamp_hi=hi, plan=plan)

        a, b = std().cpu().numpy(), prj().cpu().numpy()
        fin = np.isfinite(a) & np.isfinite(b)
        same_nan = np.array_equal(np.isnan(a), np.isnan(b))
        d = float(np.abs(a - b)[fin].max()) if fin.any() else 0.0
        line = f"{tag:10s} {prec:7s} max|sdf| {float(np.abs(a[fin]).max()):.4f}  max|diff| {d:.3e}  nan-pattern-same {same_nan}  fc0_k {plan.get('fc0_k')}"
        if time_it:
            line += f"  std {timed(std):.3f} ms  unfused {timed(lambda: std(False)):.3f} ms  proj {timed(prj):.3f} ms"
        print(line, flush=True)


if __name__ == "__main__":
    for name in ("tiny", "small", "real", "edge"):
        one(cases.build_case(name), name)
    seed = 2024
    B, N = 8, 20000
    c = {"query": synth.make_query(seed, B, N), "img_maps": synth.make_img_maps(seed, B, 224),
         "vox_maps": synth.make_vox_maps(seed, B, 128), "weights": synth.make_mlp_weights(seed),
         "trans_mat": synth.make_trans_mat(seed, B)}
    one(c, "config2", time_it=True)
    one(c, "config2-k1", time_it=True, n_kept=1, precisions=("fp16",))
    one(c, "config2-k3", time_it=True, n_kept=3, precisions=("fp16",))
