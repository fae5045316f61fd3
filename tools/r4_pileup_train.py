#!/usr/bin/env python3
"""Training step (forward + backward of the query path) with the projections of the SYNTHETIC camera and with
projections that pile onto the clamp of network/modules.py:43 the way an untrained spatial transformer's do
(bench.py --whole-model: 88 % of the points): python tools/r4_pileup_train.py [precision]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else "fp16"
    import __graft_entry__ as ge
    ge.build()
    from list_amd import hip
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    inp = bench.make_inputs("list_im2sdf_b8_n20k_224", 0, dev)
    ev = bench.HipEvents()
    out = {}
    for name, scale in (("synthetic_camera", 1.0), ("piled_on_clamp", 8.0)):
        i2 = dict(inp)
        T = inp["trans_mat"].clone()
        T[:, :3, :2] *= scale                       # u, v spread 8x wider around the map centre: most points leave the map
        i2["trans_mat"] = T
        on = float(bench._on_clamp(inp["query"], T))
        tr, _ = bench.run_train_step(prec, 6, 2, i2, hip, ev)
        out[name] = {"frac_points_on_clamp": round(on, 3), "ms_per_step": round(tr["ms_per_step"], 3),
                     "forward_query_ms": round(tr["forward_query_ms"], 3), "backward_ms": round(tr["backward_ms"], 3),
                     "kernel_ms_inline": {k: round(v, 3) for k, v in tr["kernel_ms_inline"].items()}}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
