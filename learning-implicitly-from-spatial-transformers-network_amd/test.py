#!/usr/bin/env python3
"""Inference entry point (test.py of the reference): for every test item, evaluate the SDF on the
query grid with the HIP path and extract the iso-surface.

    python test.py --model network.models.LIST --dataset datasets.Datasets.SyntheticIM2SDF -e run1 \
        --mcube_znum 256
    torchrun --nproc-per-node 8 --master-addr 127.0.0.1 test.py ...    # query axis sharded over GPUs

`--save_volume` writes the raw [res,res,res] SDF volume (npy) so that meshing can happen offline
when PyMCubes/trimesh are not installed."""
import os
import sys
import time

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_HERE))
import list_amd                                          # noqa: E402
from list_amd import arguments, utils                   # noqa: E402
from list_amd.train import wrap_model                   # noqa: E402

import numpy as np                                       # noqa: E402
import torch                                             # noqa: E402
import torch.distributed as dist                         # noqa: E402


def test_all(config, save_volume=True):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        dist.init_process_group("nccl")
    torch.cuda.set_device(local_rank if world > 1 else config.gpu)
    config.device = torch.device("cuda", torch.cuda.current_device())
    model = utils.get_class(config.model)(config).to(config.device)
    ck = config.checkpoint_dir + config.test_checkpoint
    if os.path.exists(ck):
        utils.load_model(ck, model)
        print("loaded", ck)
    else:
        print("no checkpoint at", ck, "- running with the initial weights")
    model = wrap_model(model, config).eval()
    executor = utils.get_class(config.model.replace("model", "executor"))(config, model)
    dataset = utils.get_class(config.dataset)(config, "test")
    out_dir = utils.ensure_dir(config.results_dir + "test_objs/")
    items = config.testlist or [{"cat_id": "synthetic", "shape_id": f"{i:04d}", "cam_id": i}
                                for i in range(min(len(dataset), 2))]
    rank0 = (not dist.is_initialized()) or dist.get_rank() == 0
    for it in items:
        batch = dataset.get_testdata(it["cat_id"], it["shape_id"], it["cam_id"])
        t0 = time.time()
        volume, _, _ = executor.predict_grid(batch["rgb_image"].to(config.device), batch.get("transmat"))
        torch.cuda.synchronize()
        dt = time.time() - t0
        if rank0:
            print(f"{it['cat_id']}/{it['shape_id']}: {volume.numel()} queries in {dt:.3f} s "
                  f"({volume.numel() / dt / 1e6:.2f} M points/s incl. encoders)")
            stem = utils.ensure_dir(out_dir + it["cat_id"] + "/") + f"{it['shape_id']}_{it['cam_id']}"
            if save_volume:
                np.save(stem + "_sdf.npy", volume.cpu().numpy())
            try:
                utils.generate_mesh(volume.cpu().numpy(), -0.5, 0.5, as_trimesh_obj=True).export(stem + "_pred.obj")
            except RuntimeError as e:
                print("mesh extraction skipped:", e)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    test_all(arguments.get_args())
