#!/usr/bin/env python3
"""Diagnostic: the MLP's MFMA kernel (list_gemm_nt: ping-pong and plain loop) and the vendor GEMM on fc_0's shape
(A streamed from HBM once, N = 512) and on cache-resident shapes with the same K -- separates what the schedule
costs from what the operand source costs.  Not a product path."""
import ctypes as C
import json
import sys

import torch

sys.path.insert(0, ".")
from list_amd import hip  # noqa: E402

dev = torch.device("cuda:0")
lib = hip.load()
K = 3648
SHAPES = [(160000, 512), (40192, 512), (8192, 8192), (16384, 4096), (160000 // 256 * 256, 1024)]


def timed(fn, reps=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for s, e in evs:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    return sorted(s.elapsed_time(e) for s, e in evs)[reps // 2]


rows = []
for M, N in SHAPES:
    a = (torch.randn((M, K), device=dev) * 0.5).half()
    w = (torch.randn((N, K), device=dev) * 0.02).half()
    out = torch.empty((M, N), dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def ours(flags):
        rc = lib.list_gemm_nt(a.data_ptr(), a.data_ptr(), w.data_ptr(), w.data_ptr(), None, out.data_ptr(),
                              M, N, K, flags, hip.PRECISIONS["fp16"], C.c_void_p(st))
        assert rc == 0, lib.list_last_error()

    flop = 2.0 * M * N * K
    r = {"M": M, "N": N, "K": K}
    r["pp_ms"] = timed(lambda: ours(0))
    r["plain_ms"] = timed(lambda: ours(2))
    r["vendor_ms"] = timed(lambda: torch.matmul(a, w.t()))
    ref = torch.matmul(a[:512].float(), w.float().t())
    ours(0)
    r["max_abs_vs_fp32"] = float((out[:512] - ref).abs().max())
    for k in ("pp", "plain", "vendor"):
        r[k + "_tflops"] = round(flop / r[k + "_ms"] / 1e9)
    rows.append(r)
    print(json.dumps(r), flush=True)
    del a, w, out
