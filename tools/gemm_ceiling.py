#!/usr/bin/env python3
"""Diagnostic: what the vendor GEMM libraries reach on fc_0's shape (M=160000, N=512, K=3648, fp16 -> fp16, fp32
accumulate) on this device -- a ceiling estimate for the hand-written k_gemm_nt_pp, not a product path."""
import time
import torch

dev = torch.device("cuda:0")
M, N, K = 160000, 512, 3648
for dt in (torch.float16, torch.bfloat16):
    a = torch.randn((M, K), device=dev, dtype=dt)
    w = torch.randn((N, K), device=dev, dtype=dt)
    b = torch.randn((N,), device=dev, dtype=dt)
    for name, fn in (("matmul", lambda: a @ w.t()), ("linear+relu", lambda: torch.relu(torch.nn.functional.linear(a, w, b)))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
        for s, e in evs:
            s.record(); fn(); e.record()
        torch.cuda.synchronize()
        ms = sorted(s.elapsed_time(e) for s, e in evs)[len(evs) // 2]
        print(f"{dt} {name:12s}: {ms:.3f} ms  {2 * M * N * K / ms / 1e9:.0f} TFLOP/s")
