"""CPU: `python bench.py --gpus N` with no launcher in the environment starts its N ranks itself (the shape of the
driver's N = 1 command with another N), relays rank 0's line and the launcher's exit code.  Argument / environment
plumbing only (gloo, --plumbing-check: no GPU is touched); the collectives themselves are tests/test_parallel_gloo.py
(gloo, values) and tests/test_rccl_preflight_gpu.py (one-rank RCCL group on the GPU box)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clean_env():
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK",
                        "LOCAL_WORLD_SIZE", "TORCHELASTIC_RUN_ID")}
    env["LIST_BENCH_BACKEND"] = "gloo"
    return env


def test_bench_without_a_launcher_spawns_its_ranks_and_relays_rank0():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "7", "--warmup", "2",
                        "--scaling", "strong", "--plumbing-check"], env=_clean_env(), capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "no launcher in the environment" in r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                     # rank 0 prints, rank 1 does not
    d = json.loads(lines[0])
    assert d["plumbing_check"] and d["n_gpus"] == 2 and d["scaling"] == "strong"
    assert [x["rank"] for x in d["ranks"]] == [0, 1] and [x["local_rank"] for x in d["ranks"]] == [0, 1]
    assert all(x["steps"] == 7 and x["warmup"] == 2 for x in d["ranks"])      # every rank got the caller's arguments


def test_a_failing_rank_fails_the_launcher():
    # rank 1 exits with an error before the rendezvous: torchrun tears the job down, the launcher's exit code is not 0
    env = _clean_env()
    env["LIST_BENCH_PLUMBING_FAIL_RANK"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--plumbing-check"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_self_launch_happens_before_anything_touches_the_gpu():
    """The launcher path must return before torch.cuda is used (a process that has initialised the GPU must not
    spawn the ranks from there on this pool): in the source, the self-launch precedes the first torch.cuda call."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    main = src[src.index("def main():"):]
    assert main.index("self_launch(args.gpus") < main.index("torch.cuda.")
    launcher = src[src.index("def self_launch("):src.index("def main():")]
    assert "torch.cuda" not in launcher and "exec" not in launcher.replace("sys.executable", "")
