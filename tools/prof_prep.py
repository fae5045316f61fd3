#!/usr/bin/env python3
"""Driver for `rocprofv3 --kernel-trace --stats -- python3 tools/prof_prep.py`: only the two layout hand-offs of
the metric shape, 20 times (build with LIST_HIPCC_FLAGS=-DLIST_PREP_PER_LEVEL to see the resize per level)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from list_amd import hip            # noqa: E402
from list_amd import synthetic as synth             # noqa: E402

dev = torch.device("cuda:0")
B = 8
g = torch.Generator(device=dev).manual_seed(1)
img = [torch.randn(s, generator=g, device=dev) for s in synth.img_map_shapes(B, 224)]
vox = [torch.randn(s, generator=g, device=dev) for s in synth.vox_map_shapes(B, 128)]
for md in ("f16", "f32"):
    for _ in range(10):
        hip.prep_img_maps(img, 137, md)
        hip.prep_vox_maps(vox, md)
torch.cuda.synchronize()
print("done")
