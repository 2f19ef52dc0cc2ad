"""GPU: bench.py end to end at toy scale -- the default mode R line and the sharded modes P and K with configs[3]'s read pairs,
each over a real RCCL process group of one rank (--force-dist), each with its parity block against the oracle."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--scale", "0.004", "--batch", "30000", "--steps", "2", "--warmup", "1",
           "--gather-gib", "0", "--parity-reads", "3000", "--cpu-seconds", "2", "--calibrate-scale", "0.002"] + extra
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    line = [l for l in r.stdout.split("\n") if l.startswith("{")][-1]
    return json.loads(line)


@pytest.mark.parametrize("extra", [[], ["--pairs"], ["--mode", "P", "--force-dist"], ["--mode", "P", "--pairs", "--force-dist"],
                                   ["--mode", "K", "--force-dist"], ["--mode", "K", "--pairs", "--force-dist"], ["--mode", "K", "--wire", "8", "--force-dist"], ["--config", "1", "--batch", "200000", "--genome-len", "200000"],
                                   ["--long-reads", "--batch", "3000"]])
def test_bench_line_and_parity(extra):
    res = _run(extra)
    assert res["metric"].startswith("Mreads/min") and res["value"] > 0 and res["n_gpus"] == 1
    assert res["parity"]["mismatches"] == 0 and res["parity"]["checked"] >= (200 if "--long-reads" in extra else 3000), res["parity"]
    assert "roofline" in res and res["roofline"]["frac"] > 0 and res["roofline"]["step_frac"] > 0 and "cpu_baseline" in res
    if extra == []:                                        # the default line carries the reference's speed next to the port's
        cal = res["cpu_baseline"]["reference_calibration"]
        assert cal["reference_Mreads_min"] > 0 and cal["port_Mreads_min"] > 0 and cal["reference_vs_port_mismatches"] == 0 and cal["reference_vs_gpu_mismatches"] == 0, cal
    if "--long-reads" in extra:
        assert res["config"]["Gbases_per_s"] > 0 and res["config"]["workload"].startswith("configs[4]")
    if "--config" not in extra:
        assert res["config"]["workload"].startswith("configs[")
        assert res["config"]["mode"] == (extra[extra.index("--mode") + 1] if "--mode" in extra else "R")


def test_self_launched_process_group():
    """`python bench.py --gpus N` without a launcher starts its own ranks under torch.distributed.run (here N = 1 through --force-dist):
    rank 0's JSON line is the last line of stdout, the exit code comes back"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--scale", "0.004", "--batch", "30000", "--steps", "2", "--warmup", "1",
           "--gather-gib", "0", "--parity-reads", "3000", "--cpu-seconds", "2", "--calibrate-scale", "0"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "torch.distributed.run" in r.stderr
    res = json.loads(r.stdout.strip().split("\n")[-1])
    assert res["n_gpus"] == 1 and res["parity"]["mismatches"] == 0 and res["value"] > 0
