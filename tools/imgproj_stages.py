"""Dev timing (GPU): stage times of the standard path and of the img_proj path at BASELINE config 2."""
import os
import sys

import numpy as np
import torch
import ctypes as C

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ge.build()
from list_amd import hip  # noqa: E402
from list_amd import synthetic as synth  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tools"))
from imgproj_check import dev, timed  # noqa: E402


def stage_times(fn_query, reps=8):
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(hip.N_STAGES)] for _ in range(reps)]
    out = np.zeros(hip.N_STAGES - 1)
    for r in range(reps):
        arr = (C.c_void_p * hip.N_STAGES)()
        for i, e in enumerate(evs[r]):
            e.record()                      # creates the handle
            arr[i] = e.cuda_event
        # only BEGIN, SORT, TAIL, FC0, EXACT, FC1, FC2 (no event between gathers)
        for i in range(hip.STAGE_VOX0, hip.STAGE_IMG + 1):
            arr[i] = None
        fn_query(arr)
        torch.cuda.synchronize()
        last = 0
        for i in range(1, hip.N_STAGES):
            if arr[i] is None:
                continue
            out[i - 1] += evs[r][last].elapsed_time(evs[r][i])
            last = i
    return out / reps


if __name__ == "__main__":
    prec = sys.argv[1] if len(sys.argv) > 1 else "fp16"
    seed, B, N, ms = 2024, 8, 20000, 137
    md = hip.map_dtype_for(prec)
    maps = [dev(m) for m in synth.make_img_maps(seed, B, 224)]
    voxm = [dev(m) for m in synth.make_vox_maps(seed, B, 128)]
    w = {k: dev(v) for k, v in synth.make_mlp_weights(seed).items()}
    q, T = dev(synth.make_query(seed, B, N)), dev(synth.make_trans_mat(seed, B))
    vox = hip.prep_vox_maps(voxm, md)
    packed = hip.prep_mlp_weights(w, vox.channels, 1024, prec)
    print("prep_img_maps  %.4f ms" % timed(lambda: hip.prep_img_maps(maps, ms, md), 20))
    print("prep_img_proj  %.4f ms" % timed(lambda: hip.prep_img_proj(maps, packed, ms, prec), 20))
    for k in (0, 1, 2, 3):
        print("prep_img_proj kept=%d  %.4f ms" % (k, timed(lambda: hip.prep_img_proj(maps, packed, ms, prec, n_kept_levels=k), 20)))
    img_s = hip.prep_img_maps(maps, ms, md)
    img_p = hip.prep_img_proj(maps, packed, ms, prec)
    names = hip.STAGE_NAMES
    for tag, img, fused in (("std-fused", img_s, True), ("std-unfused", img_s, False), ("proj-fused", img_p, True),
                            ("proj-unfused", img_p, False), ("std-fused", img_s, True), ("proj-fused", img_p, True)):
        t = stage_times(lambda arr: hip.sdf_query(q, T, img, vox, packed, precision=prec, stage_events=arr, fused_fc0=fused))
        tot = timed(lambda: hip.sdf_query(q, T, img, vox, packed, precision=prec, fused_fc0=fused), 20)
        print(tag, "query %.4f ms:" % tot, " ".join(f"{n}={v:.4f}" for n, v in zip(names, t) if v > 0))
