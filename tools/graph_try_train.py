"""Does a HIP graph of the whole training step of the path (layout hand-off + weight repacks + forward + forked backward + adjoint
resize) shrink what the cross-stream fork / join and the ~80 launches cost?  python tools/graph_try_train.py [precision]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import __graft_entry__ as ge
ge.build()
from list_amd import hip

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
inp = bench.make_inputs("list_im2sdf_b8_n20k_224", 0, dev)
prec = sys.argv[1] if len(sys.argv) > 1 else "fp16"
md = hip.map_dtype_for(prec)
B, N = inp["B"], inp["N"]
g0 = torch.Generator(device=dev); g0.manual_seed(4242)
gsdf = torch.randn((B, N), generator=g0, device=dev) / B
keep = {}

def step():
    img = hip.prep_img_maps(inp["img_maps"], inp["map_size"], md)
    vox = hip.prep_vox_maps(inp["vox_maps"], md)
    packed = hip.prep_mlp_weights(inp["weights"], vox.channels, img.channels, prec)
    packed_b = hip.prep_mlp_weights_bwd(inp["weights"], vox.channels, img.channels, prec)
    sdf, ctx = hip.sdf_query(inp["query"], inp["trans_mat"], img, vox, packed, precision=prec, save_for_backward=True,
                             clamp_hi=inp["clamp_hi"])
    keep["out"] = hip.sdf_query_backward(ctx, gsdf, packed_b, img_levels_like=inp["img_maps"], want_img_map=False)
    keep["sdf"] = sdf

def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

print("eager   ms/step", round(timeit(step), 4), round(timeit(step), 4))
ref = {k: v.clone() for k, v in keep["out"]["mlp"].items()}
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    step()
print("graph   ms/step", round(timeit(g.replay), 4), round(timeit(g.replay), 4))
torch.cuda.synchronize()
worst = max(float((keep["out"]["mlp"][k] - ref[k]).abs().max() / (ref[k].abs().max() + 1e-30)) for k in ref)
print("MLP gradients of the replay vs eager: worst relative difference", f"{worst:.2e}")
print("eager   ms/step", round(timeit(step), 4))
