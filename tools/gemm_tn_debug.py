import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import synth
from list_amd import hip
def dev(a): return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
for (P, M, N) in [(256, 256, 256), (512, 256, 264), (1024, 512, 3648), (4096, 256, 512)]:
    a = synth.normalish(1, (P, M)); b = synth.uniform(2, (P, N), -1, 1)
    ref = a.astype(np.float64).T @ b.astype(np.float64)
    sc = np.abs(a).astype(np.float64).T @ np.abs(b).astype(np.float64)
    for prec in ("bf16x3", "fp16", "bf16"):
        out = hip.gemm_tn(dev(a), dev(b), prec).cpu().numpy()
        print(P, M, N, prec, "rel err", float((np.abs(out - ref) / sc).max()), flush=True)
# identity check: a = [I;0] picks rows of b
P, M, N = 256, 256, 256
a = np.eye(P, M, dtype=np.float32)
b = ((np.arange(P * N, dtype=np.float32).reshape(P, N) % 251) / 16.0)
out = hip.gemm_tn(dev(a), dev(b), "bf16x3").cpu().numpy()
print("identity max diff", float(np.abs(out - b).max()))
