"""Dev (GPU, under rocprofv3 --kernel-trace): only the two 2-D preps of BASELINE config 2, 30 calls each."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as ge  # noqa: E402

ge.build()
from list_amd import hip  # noqa: E402
from list_amd import synthetic as synth  # noqa: E402
from imgproj_check import dev, timed  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "fp16"
seed, B, ms = 2024, 8, 137
md = hip.map_dtype_for(prec)
maps = [dev(m) for m in synth.make_img_maps(seed, B, 224)]
voxm = [dev(m) for m in synth.make_vox_maps(seed, B, 128)]
w = {k: dev(v) for k, v in synth.make_mlp_weights(seed).items()}
vox = hip.prep_vox_maps(voxm, md)
packed = hip.prep_mlp_weights(w, vox.channels, 1024, prec)
print("prep_img_maps  %.4f ms" % timed(lambda: hip.prep_img_maps(maps, ms, md), 30))
print("prep_img_proj  %.4f ms" % timed(lambda: hip.prep_img_proj(maps, packed, ms, prec), 30))
