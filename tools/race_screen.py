"""Race screen of the ping-pong GEMM schedule: it accumulates in the same order as the plain 2-stage loop, so the
two must agree bit for bit; any difference is an LDS-DMA ordering race.  usage: python tools/race_screen.py [reps]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from list_amd import hip
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
g = torch.Generator(device="cuda:0"); g.manual_seed(1)
bad = 0
for rep in range(reps):
    M, N, K = [(2560, 512, 3648), (160000 // 256 * 256, 512, 3648), (512, 256, 1024), (1024, 512, 4096)][rep % 4]
    a = torch.randn((M, K), generator=g, device="cuda:0")
    w = torch.randn((N, K), generator=g, device="cuda:0") * 0.05
    prec = ("fp16", "bf16", "bf16x3")[rep % 3]
    if prec == "bf16x3":      # interleaved hi / lo operands on the ping-pong schedule against the plain split kernel on planes
        x = hip.gemm_nt(a, w, None, precision=prec, interleaved=True)
        y = hip.gemm_nt(a, w, None, precision=prec)
    else:
        x = hip.gemm_nt(a, w, None, precision=prec)
        y = hip.gemm_nt(a, w, None, precision=prec, plain_loop=True)
    if not torch.equal(x, y):
        bad += 1
        print("MISMATCH", rep, M, N, K, prec, float((x - y).abs().max()))
print(f"{reps} comparisons, {bad} mismatches")
