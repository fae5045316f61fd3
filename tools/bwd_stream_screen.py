"""Hazard screen of the forked backward: with one saved forward context, the deterministic outputs (the eight MLP
gradients, the gathered perceptual-map gradient and its five encoder levels; NOT the voxel levels -- atomics, and the
gathered 32^3 level sums each cell's samples in the order its counting sort's atomic slot allocation produced) must be bit-identical
between the in-line order and the three-stream order, run after run; a stale or early read across streams would show up
as a difference.  usage: python tools/bwd_stream_screen.py [reps]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from list_amd import hip

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
inp = bench.make_inputs("list_im2sdf_b8_n20k_224", 0, dev)
bad = 0
for prec in ("fp16", "bf16x3"):
    md = hip.map_dtype_for(prec)
    img = hip.prep_img_maps(inp["img_maps"], inp["map_size"], md)
    vox = hip.prep_vox_maps(inp["vox_maps"], md)
    packed = hip.prep_mlp_weights(inp["weights"], vox.channels, img.channels, prec)
    packed_b = hip.prep_mlp_weights_bwd(inp["weights"], vox.channels, img.channels, prec)
    sdf, ctx = hip.sdf_query(inp["query"], inp["trans_mat"], img, vox, packed, precision=prec, save_for_backward=True)
    gsdf = torch.randn_like(sdf) / inp["B"]

    def run(overlap):
        o = hip.sdf_query_backward(ctx, gsdf, packed_b, overlap=overlap, img_levels_like=inp["img_maps"])
        torch.cuda.synchronize()
        det = dict(o["mlp"])
        det["img_map"] = o["img_map"]
        det.update({f"img_level{i}": t for i, t in enumerate(o["img_levels"])})
        return {k: v.clone() for k, v in det.items()}

    ref = run(False)
    for rep in range(reps):
        got = run(True)
        for k in ref:
            if not torch.equal(ref[k], got[k]):
                bad += 1
                print("MISMATCH", prec, rep, k, float((ref[k] - got[k]).abs().max()))
    print(prec, reps, "forked runs compared with the in-line order")
print("mismatches:", bad)
