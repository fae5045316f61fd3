#!/usr/bin/env python3
"""Does the forked backward modify anything it should only read?  The inputs of list_sdf_query_bwd (prepared map, prepared voxel
levels, packed weights, query, trans_mat, the forward's workspace) are cloned before the call and compared after it."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge  # noqa: E402

ge.build()
from list_amd import hip  # noqa: E402
from oracle import cases, synth  # noqa: E402
from test_hip_backward import hip_gradients, dev  # noqa: E402

hip.load()
c = cases._case(seed=909, batch=4, n=6000, img_res=64, vox_res=32)
gs = dev(synth.normalish(5, (4, 6000)))
big = cases._case(seed=8181, batch=2, n=3000, img_res=64, vox_res=128)
precision = "bf16x3"
md = hip.map_dtype_for(precision)
ref_T = None
MODE = int(sys.argv[1]) if len(sys.argv) > 1 else 0     # 0: no sync, no watch; 1: sync before the backward; 2: sync + watch
for rnd in range(3):
    hip_gradients(hip, big, synth.normalish(1, (2, 3000)), "fp16")
    hip_gradients(hip, big, synth.normalish(1, (2, 3000)), "bf16x3")
    for run in range(20):
        img_in = [dev(m) for m in c["img_maps"]]
        img = hip.prep_img_maps(img_in, 137, dtype=md)
        vox = hip.prep_vox_maps([dev(m) for m in c["vox_maps"]], dtype=md)
        params = {k: dev(v) for k, v in c["weights"].items()}
        packed = hip.prep_mlp_weights(params, vox.channels, img.channels, precision)
        packed_b = hip.prep_mlp_weights_bwd(params, vox.channels, img.channels, precision)
        q, T = dev(c["query"]), dev(c["trans_mat"])
        sdf, ctx = hip.sdf_query(q, T, img, vox, packed, precision=precision, save_for_backward=True)
        if MODE >= 1: torch.cuda.synchronize()
        watched = {"map": img.data, "query": q, "trans_mat": T, "packed": packed.data, "packed_bwd": packed_b.data if hasattr(packed_b, "data") else packed_b,
                   "fwd_workspace": ctx.keep[-3] if False else None}
        # the forward's private workspace is the uint8 tensor in ctx.keep
        for t in ctx.keep:
            if isinstance(t, torch.Tensor) and t.dtype == torch.uint8 and t.numel() > (1 << 20):
                watched["fwd_workspace"] = t
        watched = {k: v for k, v in watched.items() if isinstance(v, torch.Tensor)}
        if MODE < 2: watched = {}
        before = {k: v.clone() for k, v in watched.items()}
        out = hip.sdf_query_backward(ctx, gs, packed_b, overlap=True)
        torch.cuda.synchronize()
        Tg = out["trans_mat"].cpu().numpy()
        if ref_T is None:
            ref_T = Tg
        dev_T = float(np.abs(Tg - ref_T).max() / np.abs(ref_T).max())
        for k, v in watched.items():
            a, b = v.view(torch.uint8).reshape(-1), before[k].view(torch.uint8).reshape(-1)
            bad = (a != b).nonzero().reshape(-1)
            if bad.numel():
                print(f"[round {rnd} run {run}] {k} CHANGED during the backward: {bad.numel()} bytes, first at {int(bad[0])}, last at {int(bad[-1])} of {a.numel()}"
                      f"; d_trans_mat off by {dev_T:.2e}")
        if dev_T > 2e-6:
            print(f"[round {rnd} run {run}] d_trans_mat off by {dev_T:.2e}")
        if hasattr(hip.load(), "list_debug_trans"):
            import ctypes
            buf = (ctypes.c_uint * 16)()
            hip.load().list_debug_trans(buf, 1)
            if buf[0] or buf[4]:
                print(f"   LDS changed behind the kernel: canary mismatches {buf[0]} (first at word {buf[1]}: {buf[2]:#x}, workgroup {buf[3]}); "
                      f"point-record mismatches {buf[4]} (word {buf[5]}: {buf[6]:#x} instead of {buf[7]:#x}, thread {buf[8]}); workgroups {buf[9]}")
print("done")
