"""GPU: the sharded forms of the path on real ranks (tools/multi_gpu_selfcheck.py -- the job bench.py runs after its timed region):
modes P and K of metacache_amd/distributed.py over an RCCL process group and the C++ drivers mc_keyset_* / mc_partset_* over all devices,
every candidate against the oracle.  One GPU: a process group of one rank, the RCCL calls of the C++ drivers with a single rank
(MC_KEYSET_RCCL / MC_PARTSET_RCCL).  Two or more GPUs (skipped on the one-GPU box): one rank per device and ncclCommInitAll over all
of them -- the only difference between the two runs is the device ordinals and the number of ranks.
Reference: gpu_hashmap.cu:1253-1292, query_batch.cu:464-527, :638-652."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "multi_gpu_selfcheck.py")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(world: int, tmp_path, extra=()):
    out = str(tmp_path / "selfcheck.json")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    port = _free_port()
    env["MASTER_ADDR"] = "127.0.0.1"; env["MASTER_PORT"] = str(port)
    if world > 1:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), TOOL, "--out", out, *extra]
    else:
        cmd = [sys.executable, TOOL, "--out", out, *extra]
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert os.path.exists(out), (p.returncode, p.stdout[-2000:], p.stderr[-2000:])
    res = json.loads(open(out).read())
    assert p.returncode == 0, (res, p.stderr[-1000:])
    return res


def test_sharded_modes_single_rank(tmp_path):
    res = _run(1, tmp_path, ("--reads", "6000", "--pairs", "1500"))
    assert res["ok"] and res["ranks_seen"] == 1, res
    assert res["mode_P"] == 0 and res["mode_K"] == 0 and res["mode_T"] == 0 and res["keyset"] == 0 and res["partset"] == 0, res
    assert res["mode_K_wire"] == 4 and res["keyset_rccl"] and res["wire_bytes_per_read"] > 0, res


def test_sharded_modes_all_devices(tmp_path):
    import torch
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip("needs two or more GPUs (the driver's multi-GPU node; bench.py --gpus N runs the same job after its timed region)")
    res = _run(n, tmp_path, ("--reads", "6000", "--pairs", "1500"))
    assert res["ok"] and res["ranks_seen"] == n, res
    assert res["mode_P"] == 0 and res["mode_K"] == 0 and res["mode_T"] == 0 and res["keyset"] == 0 and res["partset"] == 0, res
    assert res["keyset_devices"] == n and res["partset_devices"] == n and res["keyset_rccl"], res
