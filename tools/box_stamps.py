"""Diagnostic: per-phase cycles of k_gather_vox_box (a -DLIST_BOX_STAMPS build given by LIST_HIP_LIB).
   LIST_HIP_LIB=variants/stamps.so python tools/box_stamps.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from list_amd import hip

dev = torch.device("cuda:0")
inp = bench.make_inputs("list_im2sdf_b8_n20k_224", 0, dev)
md = hip.map_dtype_for("fp16")
img = hip.prep_img_maps(inp["img_maps"], inp["map_size"], md)
vox = hip.prep_vox_maps(inp["vox_maps"], md)
packed = hip.prep_mlp_weights(inp["weights"], vox.channels, img.channels, "fp16")
lib = hip.load()
fn = lib.list_debug_box_stamps
fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
import numpy as np
buf = (ctypes.c_ulonglong * (2 * 4096 * 16))()
names = ["load_point", "tables+partition", "barrier", "stage box", "bucket", "scan+place", "tiles", "end barrier"]
for rep in range(3):
    hip.sdf_query(inp["query"], inp["trans_mat"], img, vox, packed, precision="fp16")
    torch.cuda.synchronize()
    assert fn(buf, 1) == 0
    if rep == 0:
        continue
    a = np.frombuffer(buf, dtype=np.uint64).reshape(2, 4096, 16).astype(np.float64)
    for lvl, label in ((1, "big box (16^3)"), (0, "small box (8^3)")):
        v = a[lvl][a[lvl][:, 12] > 0]
        wgs, runs = len(v), v[:, 8].sum()
        if wgs == 0:
            continue
        print(f"rep {rep} {label}: workgroups {wgs}, runs/wg {runs / wgs:.2f}, tiles/run {v[:, 9].sum() / runs:.1f}, "
              f"rows/run {v[:, 10].sum() / runs:.1f}, keys/run {v[:, 11].sum() / runs:.1f}")
        for i, n in enumerate(names):
            print(f"   {n:18s} mean {v[:, i].mean():9.0f}  median {np.median(v[:, i]):9.0f} cycles per workgroup")
        print(f"   total              {v[:, :8].sum(1).mean():9.0f}")
