"""Sharding of the (batch x query-point) axis: one process per GPU, torch.distributed.

The reference's only multi-GPU mechanism is single-process nn.DataParallel (train.py:126), which
scatters the batch, re-broadcasts 105 M parameters every forward and gathers every output on GPU 0.
Here every query point is independent given its image's maps, so:

  * training / batched queries : the BATCH axis is partitioned (B_global = world * B_local);
  * inference on one image     : the QUERY axis is partitioned (contiguous slices of the grid);
  * the only data-path exchange is ONE all-gather of the fp32 SDF shards ([B_local,N] per rank,
    640 KB at B_local=8, N=20k) so that every rank can evaluate the reference's full-batch SDFLoss
    (network/losses.py:21-22).  Backend "nccl" is RCCL over xGMI on ROCm; the CPU tests use gloo.
"""
import torch
import torch.distributed as dist


def world_info(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def _local_only(world, group=None):
    """True when an exchange is a no-op: one process.  LIST_FORCE_COLLECTIVES=1 sends a one-rank process group down
    the collective branches all the same -- the pre-flight of the RCCL code paths on a one-GPU box
    (tests/test_rccl_preflight_gpu.py): the calls, dtypes, layouts and the async work handle are those of N ranks."""
    import os
    if world > 1:
        return False
    return not (os.environ.get("LIST_FORCE_COLLECTIVES", "0") == "1" and dist.is_available() and dist.is_initialized())


def shard_range(total, rank, world):
    """Contiguous, balanced [begin, end) of `total` items for `rank` (first `total % world` ranks
    get one extra item)."""
    base, extra = divmod(total, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def gather_sdf_shards(sdf_local, out=None, group=None, async_op=False):
    """All-gather equally-shaped SDF shards along dim 0 -> [world*B_local, N].

    async_op=True (RCCL only) returns (out, work): the collective runs on RCCL's stream behind the work already
    queued on the current stream, and the caller's next kernels do not wait for it; `work.wait()` orders the
    current stream after it (needed before `out` is read or `sdf_local` is overwritten).  Other backends and a
    single process complete in place and return (out, None)."""
    rank, world = world_info(group)
    if _local_only(world, group):
        if out is None:
            out = sdf_local
        else:
            out.copy_(sdf_local)
        return (out, None) if async_op else out
    sdf_local = sdf_local.contiguous()
    if out is None:
        out = torch.empty((world * sdf_local.shape[0],) + tuple(sdf_local.shape[1:]),
                          dtype=sdf_local.dtype, device=sdf_local.device)
    if sdf_local.is_cuda and dist.get_backend(group) != "nccl":
        # CPU-side backends (gloo rehearsals): stage through the host
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(host, sdf_local.cpu(), group=group)
        out.copy_(host)
        return (out, None) if async_op else out
    if async_op and sdf_local.is_cuda:
        return out, dist.all_gather_into_tensor(out, sdf_local, group=group, async_op=True)
    dist.all_gather_into_tensor(out, sdf_local, group=group)
    return (out, None) if async_op else out


def gather_batch_ragged(sdf_local, batch_global, group=None):
    """All-gather of BATCH-axis shards produced with shard_range when the global batch does not divide by the ranks
    (strong scaling, BASELINE config 3 split over 1 / 2 / 4 / 8 ranks and anything in between): every rank sends a
    buffer of ceil(batch_global / world) images of which it fills its own, the result is trimmed to
    [batch_global, N] in rank order.  The same padded exchange bench.py --scaling strong keeps in its timed step."""
    rank, world = world_info(group)
    if _local_only(world, group):
        return sdf_local
    b_pad = -(-batch_global // world)
    b, e = shard_range(batch_global, rank, world)
    if sdf_local.shape[0] != e - b:
        raise RuntimeError(f"rank {rank} owns images [{b}, {e}) of {batch_global}, got {sdf_local.shape[0]}")
    buf = torch.zeros((b_pad,) + tuple(sdf_local.shape[1:]), dtype=sdf_local.dtype, device=sdf_local.device)
    buf[: e - b] = sdf_local
    allb = gather_sdf_shards(buf, group=group)
    parts = []
    for r in range(world):
        rb, re = shard_range(batch_global, r, world)
        parts.append(allb[r * b_pad: r * b_pad + (re - rb)])
    return torch.cat(parts)


def gather_ragged_points(values_local, total, group=None):
    """All-gather of ragged 1-D shards produced with shard_range (query-axis partition of one
    image's grid): pads to the largest shard, gathers, and trims -> [total]."""
    rank, world = world_info(group)
    if _local_only(world, group):
        return values_local
    longest = (total + world - 1) // world
    pad = torch.zeros((longest,), dtype=values_local.dtype, device=values_local.device)
    pad[: values_local.numel()] = values_local.reshape(-1)
    buf = torch.empty((world * longest,), dtype=values_local.dtype, device=values_local.device)
    dist.all_gather_into_tensor(buf, pad, group=group)
    parts = []
    for r in range(world):
        b, e = shard_range(total, r, world)
        parts.append(buf[r * longest: r * longest + (e - b)])
    return torch.cat(parts)


def full_batch_sdf_loss(sdf_local, targets_local, sdf_scale=1.0, group=None):
    """The reference's SDFLoss value (network/losses.py:21-22: mean over the batch of the
    per-image sum of squared errors) evaluated over the GLOBAL batch on every rank."""
    pred = gather_sdf_shards(sdf_local, group=group)
    tgt = gather_sdf_shards(targets_local, group=group)
    return torch.mean(((tgt * sdf_scale - pred) ** 2).sum(-1))


def full_batch_value(local, full):
    """`local` for the gradient, `full` for the value: a scalar whose forward value is the full-batch number
    (identical on every rank) while autograd sees only this rank's term.  With equal shards DistributedDataParallel's
    gradient average over the ranks is then exactly the gradient of the full-batch loss."""
    return local + (full.detach() - local.detach())


def all_reduce_mean(value, group=None):
    """Mean over the ranks of a (detached) scalar / small tensor."""
    rank, world = world_info(group)
    if _local_only(world, group):
        return value.detach()
    v = value.detach().clone()
    if v.is_cuda and dist.get_backend(group) != "nccl":
        host = v.cpu()
        dist.all_reduce(host, group=group)
        return (host / world).to(v.device)
    dist.all_reduce(v, group=group)
    return v / world


def broadcast_from_rank0(tensors, group=None):
    """Overwrite every tensor in `tensors` (same shapes on all ranks) with rank 0's values, in place.  Used by the
    sharded inference: every rank encodes the image (for shapes, dtypes and memory formats), then takes rank 0's
    maps, so that the sharded SDF grid is the unsharded one bit for bit even though MIOpen's convolutions are not
    run-to-run deterministic."""
    rank, world = world_info(group)
    if _local_only(world, group):
        return tensors
    for t in tensors:
        if t.is_cuda and dist.get_backend(group) != "nccl":
            host = t.detach().cpu().contiguous()
            dist.broadcast(host, src=0, group=group)
            t.detach().copy_(host)
        elif t.is_contiguous():
            dist.broadcast(t.detach(), src=0, group=group)
        else:                                   # channels-last maps: broadcast the dense storage order
            flat = t.detach().permute(*_storage_order(t)).contiguous()
            dist.broadcast(flat, src=0, group=group)
            t.detach().permute(*_storage_order(t)).copy_(flat)
    return tensors


def _storage_order(t):
    """Dimension order from the largest stride to the smallest (the permutation that makes `t` contiguous)."""
    return sorted(range(t.dim()), key=lambda d: (-t.stride(d), d))
