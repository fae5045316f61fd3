#!/usr/bin/env python3
"""Run-to-run spread of every gradient of the case of tests/test_hip_backward.py::test_backward_large_batch_statistics in the forked
backward, after other work has moved the allocator (where d_trans_mat was seen to deviate by 1e-4): python tools/trans_noise_probe.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge  # noqa: E402

ge.build()
from list_amd import hip  # noqa: E402
from oracle import cases, synth  # noqa: E402
from test_hip_backward import hip_gradients  # noqa: E402

hip.load()
c = cases._case(seed=909, batch=4, n=6000, img_res=64, vox_res=32)
gs = synth.normalish(5, (4, 6000))
big = cases._case(seed=8181, batch=2, n=3000, img_res=64, vox_res=128)
want = {} if len(sys.argv) < 2 else eval(sys.argv[1])
for rnd in range(3):
    hip_gradients(hip, big, synth.normalish(1, (2, 3000)), "fp16")
    hip_gradients(hip, big, synth.normalish(1, (2, 3000)), "bf16x3")
    outs = [hip_gradients(hip, c, gs, "bf16x3", want=dict(want, overlap=True))[1] for _ in range(10)]
    for k in sorted(outs[0]):
        st = np.stack([o[k] for o in outs]).astype(np.float64)
        med = np.median(st, axis=0)
        m = np.abs(med).max() + 1e-30
        dev = np.abs(st - med).reshape(len(outs), -1).max(axis=1) / m
        if dev.max() > 2e-6:
            r = int(dev.argmax())
            idx = np.unravel_index(np.abs(st[r] - med).argmax(), med.shape)
            print(f"[round {rnd}] {k}: run {r} deviates by {dev.max():.2e} of the largest entry at {idx} "
                  f"(value {st[r][idx]:.6g}, median {med[idx]:.6g}); other runs <= {np.sort(dev)[-2]:.2e}")
            if k == "d_trans_mat":
                print("   difference:\n", (st[r] - med))
print("done")
