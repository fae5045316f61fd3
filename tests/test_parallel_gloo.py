"""CPU, world_size 2, gloo: the sharding helpers of list_amd.parallel (the N>1 path of bench.py)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from list_amd import parallel as P
        # batch-axis shards: global batch 6 = 2 ranks x 3 images, N = 17
        g = torch.Generator().manual_seed(0)
        full_pred = torch.randn((6, 17), generator=g)
        full_tgt = torch.randn((6, 17), generator=g)
        b, e = P.shard_range(6, rank, world)
        gathered = P.gather_sdf_shards(full_pred[b:e].clone())
        assert torch.equal(gathered, full_pred)                       # bit-for-bit
        # the overlapped form bench.py uses (a work handle with RCCL, completed in place elsewhere)
        out = torch.empty_like(full_pred)
        got, work = P.gather_sdf_shards(full_pred[b:e].clone(), out=out, async_op=True)
        if work is not None:
            work.wait()
        assert got is out and torch.equal(out, full_pred)
        loss = P.full_batch_sdf_loss(full_pred[b:e].clone(), full_tgt[b:e].clone(), 2.0)
        ref = torch.mean(((full_tgt * 2.0 - full_pred) ** 2).sum(-1))
        assert torch.equal(loss, ref)
        # query-axis shards of one image's grid, ragged (37 points over 2 ranks: 19 + 18)
        grid_vals = torch.arange(37, dtype=torch.float32) * 0.5
        b, e = P.shard_range(37, rank, world)
        assert (e - b) == (19 if rank == 0 else 18)
        whole = P.gather_ragged_points(grid_vals[b:e].clone(), 37)
        assert torch.equal(whole, grid_vals)
        np.save(os.path.join(out_dir, f"ok_{rank}.npy"), np.array([1]))
    finally:
        dist.destroy_process_group()


def test_shard_range_partitions_exactly():
    from list_amd import parallel as P
    for total in (1, 7, 8, 20000, 2097152):
        for world in (1, 2, 3, 8):
            spans = [P.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1


def test_gather_world2_gloo(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert all(os.path.exists(tmp_path / f"ok_{r}.npy") for r in range(2))


def test_single_process_passthrough():
    from list_amd import parallel as P
    x = torch.arange(6.0).reshape(2, 3)
    assert P.gather_sdf_shards(x) is x
    got, work = P.gather_sdf_shards(x, async_op=True)
    assert got is x and work is None
    assert P.world_info() == (0, 1)
