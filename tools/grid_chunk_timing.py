#!/usr/bin/env python3
"""Diagnostic: executors.LIST.predict_grid on a 256^3 grid (random-init model, per-image stage frozen) for several
query chunk sizes (the reference's --test_pointnum, default 65536)."""
import sys
import time

import torch

sys.path.insert(0, ".")
from list_amd import arguments, utils          # noqa: E402
from list_amd.train import _Module              # noqa: E402
from list_amd import synthetic as synth                        # noqa: E402

dev = torch.device("cuda:0")
res = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for prec in ("fp16", "bf16x3"):
    for chunk in (65536, 262144, 1048576, 4194304):
        cfg = arguments.default_config(vox_res=128, train_batch_size=1, mcube_znum=res, test_pointnum=chunk)
        cfg.device = dev
        cfg.precision = prec
        net = utils.get_class("network.models.LIST")(cfg).to(dev).eval()
        ex = utils.get_class("network.executors.LIST")(cfg, _Module(net))
        img = torch.from_numpy(synth.uniform(78, (1, 3, 224, 224))).to(dev)
        with torch.no_grad():
            enc = net.encode(img)
        net.encode = lambda *a, **k: enc       # only the query path is timed
        for it in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            vol, _, _ = ex.predict_grid(img, shard=False)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"{prec} chunk {chunk:8d}: {dt * 1e3:8.1f} ms  {res ** 3 / dt / 1e6:7.1f} M points/s", flush=True)
        del net, ex
