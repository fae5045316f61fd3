"""Child process of tests/test_rccl_preflight_gpu.py: a ONE-rank `nccl` (= RCCL) process group on cuda:0 driving the
collective branches of list_amd.parallel (LIST_FORCE_COLLECTIVES=1) -- the code the 8-GPU run executes, with its
device tensors, dtypes, layouts and the asynchronous work handle bench.py's exchange relies on.  Prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", str(29400 + os.getpid() % 500))
os.environ["LIST_FORCE_COLLECTIVES"] = "1"
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from list_amd import parallel as P  # noqa: E402

res = {"backend": dist.get_backend(), "world": dist.get_world_size()}
g = torch.Generator(device=dev).manual_seed(7)
x = torch.randn((8, 20000), generator=g, device=dev)

# gather_sdf_shards: in line, into a caller's buffer, and with the async work handle (RCCL's stream beside the
# caller's next kernels; wait() orders the current stream behind it)
res["gather_inline"] = bool(torch.equal(P.gather_sdf_shards(x), x))
out = torch.full_like(x, float("nan"))
got, work = P.gather_sdf_shards(x, out=out, async_op=True)
y = (x * 2).sum()                       # the caller's next kernel does not wait for the exchange
res["async_handle"] = type(work).__name__ if work is not None else None
if work is not None:
    work.wait()
torch.cuda.synchronize()
res["gather_async"] = bool(got is out and torch.equal(out, x)) and bool(torch.isfinite(y))

# ragged batch shards (strong scaling) and ragged query shards (one image's grid)
res["gather_batch_ragged"] = bool(torch.equal(P.gather_batch_ragged(x[:5], 5), x[:5]))
v = torch.randn((12345,), generator=g, device=dev)
res["gather_ragged_points"] = bool(torch.equal(P.gather_ragged_points(v, 12345), v))

# loss exchange (the reference's full-batch SDFLoss on every rank) and the scalar reduction
tgt = torch.randn((8, 20000), generator=g, device=dev)
want = torch.mean(((tgt - x) ** 2).sum(-1))
res["full_batch_sdf_loss"] = bool(torch.allclose(P.full_batch_sdf_loss(x, tgt), want, rtol=1e-6))
res["all_reduce_mean"] = bool(torch.equal(P.all_reduce_mean(want), want))

# broadcast of rank 0's maps: dense and channels-last storage
a = torch.randn((2, 16, 8, 8, 8), generator=g, device=dev)
b = torch.randn((2, 64, 14, 14), generator=g, device=dev).contiguous(memory_format=torch.channels_last)
a0, b0 = a.clone(), b.clone()
P.broadcast_from_rank0([a, b])
res["broadcast"] = bool(torch.equal(a, a0) and torch.equal(b, b0) and b.is_contiguous(memory_format=torch.channels_last))

# the rank census bench.py prints (`rccl_ranks_observed`)
mine = torch.tensor([0.0, 1.5, 8.0], dtype=torch.float64, device=dev)
allr = torch.empty((3,), dtype=torch.float64, device=dev)
dist.all_gather_into_tensor(allr, mine)
res["census"] = allr.cpu().tolist()
dist.barrier()
dist.destroy_process_group()
print(json.dumps(res))
