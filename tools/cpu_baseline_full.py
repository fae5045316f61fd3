"""What bench.py's bounded CPU sample (2 of 8 images) is worth: the same CPU baseline (oracle/torch_ops.py = the reference's
torch op sequence, fp32, no_grad, all host threads) on 1, 2, 4 and ALL 8 images of the metric workload.
    python tools/cpu_baseline_full.py > profiles/rNN_cpu_baseline_8_images.json      (on the GPU box's host; no GPU needed)"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from list_amd import synthetic as synth          # noqa: E402
from oracle import torch_ops as TO               # noqa: E402  (the CPU baseline IS the oracle: test infrastructure)

B, N, img_res, vox_res = 8, 20000, 224, 128
g = torch.Generator().manual_seed(333)
img = [torch.randn(s, generator=g) for s in synth.img_map_shapes(B, img_res)]
vs = synth.vox_map_shapes(B, vox_res)
vox = [torch.rand(vs[0], generator=g)] + [torch.randn(s, generator=g) for s in vs[1:]]
q = torch.rand((B, N, 3), generator=g) - 0.5
T = torch.from_numpy(synth.make_trans_mat(333, B))
w = {k: torch.from_numpy(v) for k, v in synth.make_mlp_weights(333).items()}
out = {"workload": "list_im2sdf_b8_n20k_224", "cores": torch.get_num_threads(), "os_cpu_count": os.cpu_count(),
       "torch": torch.__version__, "kind": "port", "runs": []}
for ns in (1, 2, 4, 8):
    args = (q[:ns], [m[:ns] for m in img], [m[:ns] for m in vox], T[:ns], w)
    TO.list_query(*args)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        TO.list_query(*args)
        ts.append(time.perf_counter() - t0)
    med = sorted(ts)[1]
    out["runs"].append({"images": ns, "points": ns * N, "median_s": med, "points_per_s": ns * N / med})
    print(f"{ns} images: {ns * N / med:,.0f} points/s ({med:.2f} s)", file=sys.stderr)
print(json.dumps(out, indent=1))
