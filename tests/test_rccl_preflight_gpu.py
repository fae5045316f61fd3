"""GPU: pre-flight of the RCCL code paths on the one-GPU box.  The 8-GPU scaling run is the driver's; what can be
checked here is that every `nccl` branch of list_amd.parallel and the asynchronous exchange of bench.py's step run
on device tensors through a real RCCL communicator (a one-rank group) and return the right values.  Not a scaling
number.  Reference strategy replaced: nn.DataParallel, /root/reference train.py:126."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_one_rank_rccl_group_drives_every_collective_branch():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_child_rccl_preflight.py")], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["backend"] == "nccl" and d["world"] == 1
    assert d["async_handle"] is not None, "RCCL returned no work handle for async_op=True"
    for k in ("gather_inline", "gather_async", "gather_batch_ragged", "gather_ragged_points", "full_batch_sdf_loss",
              "all_reduce_mean", "broadcast"):
        assert d[k] is True, (k, d)
    assert d["census"] == [0.0, 1.5, 8.0]


@pytest.mark.gpu
def test_bench_step_with_the_exchange_on_a_one_rank_rccl_group():
    """bench.py's N > 1 step (query kernels + asynchronous all-gather on RCCL's stream, two alternating buffers,
    exchange check) as the driver launches it, with one rank: WORLD_SIZE = 1 under torch.distributed.run and
    LIST_BENCH_FORCE_EXCHANGE=1."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(LIST_BENCH_FORCE_EXCHANGE="1", LIST_FORCE_COLLECTIVES="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2",
                        "--precision", "fp16", "--no-cpu-baseline", "--sustained-steps", "0"], env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["ranks"]["backend"] == "nccl" and d["ranks"]["rccl_ranks_observed"] == 1
    chk = d["ranks"]["exchange_check"]
    assert chk["own_shard_at_own_offset"] and chk["all_finite"] and chk["every_rank_delivered"], chk
    assert "asynchronous all-gather unavailable" not in r.stderr
    assert d["value"] > 1e7
