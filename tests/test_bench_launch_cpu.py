"""CPU: `python bench.py --gpus N` (the form of the driver's recorded command) starts its own N ranks under torch.distributed.run
instead of refusing at argument parsing; without GPUs the run then fails INSIDE the ranks, and the exit code comes back."""
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench_module():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_self_launch_command_is_the_drivers_form():
    m = _bench_module()
    cmd = m.self_launch_command(["--gpus", "8", "--steps", "20", "--warmup", "5"], 8, 29577)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29577"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]        # the same arguments, unchanged


def test_plain_gpus_2_fails_inside_the_ranks_not_at_argument_parsing():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=600)
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        return                                                   # (a box with two GPUs: the run itself is test_gpu_bench_modes' business)
    assert r.returncode != 0
    assert "torch.distributed.run" in r.stderr                   # the launch line bench.py prints
    assert "must be launched" not in r.stderr and "ChildFailedError" in r.stderr, r.stderr[-2000:]
