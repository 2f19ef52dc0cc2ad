"""Experiment (round 5): does a lookup-shaped kernel of 32 registers and no LDS run BESIDE gw_filter_count_kernel -- in what five filter waves
per SIMD leave of a CU -- without taking the filter's time?  Full-scale table, four batches of 5 x 10^6 reads one after the other on the
context's stream, with and without a side kernel on a second stream that does the same number of random table-sized reads the batches'
lookups do (tools/gather_peak.hip: mcg_side_launch).    python tools/coresidency_probe.py --scale 1"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import bench  # noqa: E402
from metacache_amd import build, synthdb  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--batch", type=int, default=5_000_000)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--side-gib", type=float, default=32.0)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    lib = ctypes.CDLL(build.build_gather_peak())
    lib.mcg_side_launch.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p]
    c2 = dict(bench.CFG2); c2["genera"] = max(2, int(round(c2["genera"] * args.scale)))
    spec = synthdb.phylogeny(**c2)
    shards = max(1, int(np.ceil(spec.total_bases // 112 * 16 / 1.4e9)))
    db, info = synthdb.build_database(spec, shards=shards, max_candidates=2, max_load_factor=0.3, report=lambda m: print(m, file=sys.stderr, flush=True))
    B = args.batch
    gen = synthdb.GpuSynth(0)
    P = synthdb.read_params(spec, 3100)
    batches = []
    for s in range(2):
        t = torch.zeros(B * bench.PAD_LEN + 16, dtype=torch.uint8, device=dev)
        gen.reads(spec, P, s * B, B, t)
        batches.append(t)
    qinfo = torch.zeros((B, 4), dtype=torch.int32, device=dev)
    qinfo[:, 0] = torch.arange(B, device=dev, dtype=torch.int32) * bench.PAD_LEN
    qinfo[:, 1] = bench.READ_LEN; qinfo[:, 2] = qinfo[:, 0]
    side_bytes = int(args.side_gib * (1 << 30))
    side = torch.ones(side_bytes // 8, dtype=torch.int64, device=dev)
    sink = torch.zeros(1 << 24, dtype=torch.int32, device=dev)
    st2 = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()

    def steps():
        for i in range(args.steps):
            db.query_device(batches[i % 2].data_ptr(), qinfo.data_ptr(), B, B * bench.PAD_LEN, max_win_uniform=3)
        db.synchronize()

    steps()
    res = []
    lookups = args.steps * B * 40                              # what the batches' own lookup kernels request
    for lanes, blocks_per_cu in ((0, 0), (4, 4), (4, 8), (4, 16), (1, 4), (1, 8), (1, 16)):
        ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if lanes:
            blocks = 256 * blocks_per_cu
            iters = max(1, int(lookups * lanes / (blocks * 64 * 4)))
            ea.record(st2)
            rc = lib.mcg_side_launch(side.data_ptr(), side_bytes, lanes, blocks, iters, sink.data_ptr(), st2.cuda_stream)
            assert rc == 0
            eb.record(st2)
        steps()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        run = {"side_lanes_per_unit": lanes, "side_blocks_per_cu": blocks_per_cu, "wall_ms": round(wall * 1e3, 2), "ms_per_step": round(wall * 1e3 / args.steps, 3),
               "side_kernel_ms": round(ea.elapsed_time(eb), 2) if lanes else None, "side_units": lookups if lanes else 0}
        print(json.dumps(run), flush=True)
        res.append(run)
    # the side kernels alone
    for lanes, blocks_per_cu in ((4, 8), (1, 8)):
        ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        blocks = 256 * blocks_per_cu
        iters = max(1, int(lookups * lanes / (blocks * 64 * 4)))
        ea.record(st2)
        lib.mcg_side_launch(side.data_ptr(), side_bytes, lanes, blocks, iters, sink.data_ptr(), st2.cuda_stream)
        eb.record(st2)
        torch.cuda.synchronize()
        print(json.dumps({"side_alone_lanes": lanes, "blocks_per_cu": blocks_per_cu, "ms": round(ea.elapsed_time(eb), 2), "G_units_per_s": round(lookups / ea.elapsed_time(eb) / 1e6, 1)}), flush=True)
    db.close()


if __name__ == "__main__":
    main()
