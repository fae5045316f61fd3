import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import sys; sys.path.insert(0, os.getcwd())
from list_amd import parallel as P
x = torch.randn(8, 20000, device="cuda")
# world_size 1 through the real RCCL work-handle path
out = torch.empty_like(x)
w = dist.all_gather_into_tensor(out, x, async_op=True)
y = x * 2          # next kernels do not wait
w.wait(); torch.cuda.synchronize()
print("rccl async all_gather world=1 ok", bool(torch.equal(out, x)), type(w).__name__)
dist.barrier(); dist.destroy_process_group()
