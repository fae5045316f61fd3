"""Does a HIP graph of the forward step (prep + query) shrink the inter-kernel gaps?  Scratch measurement."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from list_amd import hip

dev = torch.device("cuda:0")
inp = bench.make_inputs("list_im2sdf_b8_n20k_224", 0, dev)
prec = "fp16"
md = hip.map_dtype_for(prec)
sdf = torch.empty((inp["B"], inp["N"]), device=dev)

def step():
    img = hip.prep_img_maps(inp["img_maps"], inp["map_size"], md)
    vox = hip.prep_vox_maps(inp["vox_maps"], md)
    packed = hip.prep_mlp_weights(inp["weights"], vox.channels, img.channels, prec)
    hip.sdf_query(inp["query"], inp["trans_mat"], img, vox, packed, precision=prec, out=sdf, clamp_hi=inp["clamp_hi"])

def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

print("eager   ms/step", round(timeit(step), 4))
ref = sdf.clone()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    step()
print("graph   ms/step", round(timeit(g.replay), 4))
print("same result", bool(torch.equal(ref, sdf)))
