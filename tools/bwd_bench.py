"""Per-kernel timing of the training step of the path (forward + backward) at the bench workload.
usage: python tools/bwd_bench.py [precision] [steps]"""
import ctypes
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                           # noqa: E402
from list_amd import hip                               # noqa: E402

precision = sys.argv[1] if len(sys.argv) > 1 else "fp16"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
device = torch.device("cuda:0")
inp = bench.make_inputs("list_im2sdf_b8_n20k_224", 0, device)
ev = bench.HipEvents()
md = hip.map_dtype_for(precision)
gsdf = torch.randn((inp["B"], inp["N"]), device=device) / inp["B"]
acc = np.zeros(hip.N_BWD_STAGES - 1)
extra = np.zeros(4)
total = 0.0
for it in range(steps + 2):
    arr = (ctypes.c_void_p * hip.N_BWD_STAGES)(*[ev.create() for _ in range(hip.N_BWD_STAGES)])
    e = [ev.create() for _ in range(4)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    img = hip.prep_img_maps(inp["img_maps"], inp["map_size"], md)
    vox = hip.prep_vox_maps(inp["vox_maps"], md)
    packed = hip.prep_mlp_weights(inp["weights"], vox.channels, img.channels, precision)
    ev.record(e[0])
    packed_b = hip.prep_mlp_weights_bwd(inp["weights"], vox.channels, img.channels, precision)
    ev.record(e[1])
    sdf, ctx = hip.sdf_query(inp["query"], inp["trans_mat"], img, vox, packed, precision=precision,
                             save_for_backward=True)
    ev.record(e[2])
    in_call = os.environ.get("LIST_BWD_LEVELS_IN_CALL", "1") == "1"
    out = hip.sdf_query_backward(ctx, gsdf, packed_b, stage_events=arr,
                                 overlap=os.environ.get("LIST_BWD_OVERLAP", "1") == "1",
                                 img_levels_like=inp["img_maps"] if in_call else None,
                                 want_img_map=os.environ.get("LIST_BWD_WANT_IMG_MAP", "0") == "1")
    ev.record(e[3])
    lv = out["img_levels"] if in_call else hip.img_map_grad_to_levels(out["img_map"], inp["img_maps"])
    e4 = ev.create(); ev.record(e4)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if it >= 2:
        total += dt
        for s in range(hip.N_BWD_STAGES - 1):
            acc[s] += ev.elapsed_ms(ctypes.c_void_p(arr[s]), ctypes.c_void_p(arr[s + 1]))
        extra += [ev.elapsed_ms(e[0], e[1]), ev.elapsed_ms(e[1], e[2]), ev.elapsed_ms(e[3], e4), ev.elapsed_ms(e[2], e[3])]
    del out, ctx, lv
print(f"precision {precision}: fwd+bwd wall {1e3 * total / steps:.3f} ms/step")
print(f"  prep_weights_bwd {extra[0] / steps:.3f}  forward(query only) {extra[1] / steps:.3f}  img_grad_to_levels {extra[2] / steps:.3f}")
for n, v in zip(hip.BWD_STAGE_NAMES, acc / steps):
    print(f"  {n:16s} {v:.3f} ms")
print(f"  backward (stream time, stages overlap)   {extra[3] / steps:.3f} ms")
