"""Child process of tests/test_fused_fc0_gpu.py: the SDF of a fixed list of fp16 queries, written to an .npz.  The parent
runs it with LIST_FUSED_FC0 unset (fc_0 with the perceptual block produced on chip), =0 (2-D gather kernel +
k_gemm_nt_pp) and =x (the 128 x 512 tile with every K-tile from X) -- the environment is read when the library makes its
first query, hence a fresh process each -- and compares the files bit for bit."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import cases                     # noqa: E402  (inputs of the parity cases: test infrastructure)
from list_amd import synthetic as synth      # noqa: E402


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


def run(hip, c, res, tag, map_size=137, clamp_hi=136.0, sorts=(True, False), train=False, precision="fp16"):
    md = hip.map_dtype_for(precision)
    img = hip.prep_img_maps([dev(m) for m in c["img_maps"]], map_size, md)
    vox = hip.prep_vox_maps([dev(m) for m in c["vox_maps"]], md)
    packed = hip.prep_mlp_weights({k: dev(v) for k, v in c["weights"].items()}, vox.channels, img.channels, precision)
    q, T = dev(c["query"]), dev(c["trans_mat"])
    tag = tag if precision == "fp16" else f"{tag}_{precision}"
    for sort in sorts:
        plan = {}
        sdf = hip.sdf_query(q, T, img, vox, packed, precision=precision, sort_points=sort, clamp_hi=clamp_hi, plan=plan)
        res[f"{tag}_{'sorted' if sort else 'unsorted'}"] = sdf.cpu().numpy()
        res[f"{tag}_fused_fc0"] = np.int32(plan["fused_fc0"])
    if train:       # a forward that keeps its activations (two-launch tail, H1 / H2 / H3 in the workspace)
        plan = {}
        sdf, _ = hip.sdf_query(q, T, img, vox, packed, precision=precision, clamp_hi=clamp_hi, save_for_backward=True, plan=plan)
        res[f"{tag}_train"] = sdf.cpu().numpy()
        res[f"{tag}_train_plan_fused_fc0"] = np.int32(plan["fused_fc0"])


def main(out_path, full):
    import __graft_entry__ as ge
    ge.build()
    from list_amd import hip
    res = {}
    for name in ("tiny", "small", "real", "edge") + tuple(cases.NONFINITE_CASE_NAMES):
        run(hip, cases.build_case(name), res, name, train=(name == "small"))
        run(hip, cases.build_case(name), res, name, train=(name == "small"), precision="bf16x3")
    run(hip, cases.build_case("small"), res, "small", precision="bf16")
    if full:
        seed = 2024
        for tag, B, N, img_res, ms, hi in (("config2", 8, 20000, 224, 137, 136.0), ("config5", 8, 50000, 512, 274, 273.0)):
            c = {"query": synth.make_query(seed, B, N), "img_maps": synth.make_img_maps(seed, B, img_res),
                 "vox_maps": synth.make_vox_maps(seed, B, 128), "weights": synth.make_mlp_weights(seed),
                 "trans_mat": synth.make_trans_mat(seed, B) * (np.array([[[ms / 137.0, ms / 137.0, 1.0]]], np.float32))}
            run(hip, c, res, tag, ms, hi, sorts=(True,))
            if tag == "config2":
                run(hip, c, res, tag, ms, hi, sorts=(True,), precision="bf16x3")
    np.savez(out_path, **res)


if __name__ == "__main__":
    main(sys.argv[1], len(sys.argv) > 2 and sys.argv[2] == "full")
