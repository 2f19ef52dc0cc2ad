"""Builds the configs[2] table once (bench.py's collection at --scale) and times the hot path for settings of run-time tuning switches
(mc_set_tuning), e.g. the filter and the counting apart or the grids' blocks per CU:
  python tools/tune_gw.py --scale 1 --set gw_fuse=0,1      python tools/tune_gw.py --scale 1 --set filter_bpc=5,10,20      --set count_bpc=8,16,24"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from metacache_amd import synthdb  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--batch", type=int, default=5_000_000)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--set", default="gw_fuse=1", help="name=v1,v2,...: one timed run per value")
    ap.add_argument("--fixed", default="", help="name=value[,name=value ...]: switches set once, before the runs")
    ap.add_argument("--load-factor", type=float, default=0.3)
    ap.add_argument("--out", default="")
    ap.add_argument("--pipes", default="1", help="1,2: batches in flight (2 = mc_query_device(MC_DEFER_TAIL) alternating between the two pipes)")
    ap.add_argument("--nbatches", type=int, default=2, help="distinct batches resident")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    c2 = dict(bench.CFG2); c2["genera"] = max(2, int(round(c2["genera"] * args.scale)))
    spec = synthdb.phylogeny(**c2)
    shards = max(1, int(np.ceil(spec.total_bases // 112 * 16 / 1.4e9)))
    db, info = synthdb.build_database(spec, shards=shards, max_candidates=2, max_load_factor=args.load_factor,
                                      report=lambda m: print(m, file=sys.stderr, flush=True))
    B = args.batch
    gen = synthdb.GpuSynth(0)
    P = synthdb.read_params(spec, 3100)
    batches = []
    for s in range(args.nbatches):
        t = torch.zeros(B * bench.PAD_LEN + 16, dtype=torch.uint8, device=dev)
        gen.reads(spec, P, s * B, B, t)
        batches.append(t)
    qinfo = torch.zeros((B, 4), dtype=torch.int32, device=dev)
    qinfo[:, 0] = torch.arange(B, device=dev, dtype=torch.int32) * bench.PAD_LEN
    qinfo[:, 1] = bench.READ_LEN; qinfo[:, 2] = qinfo[:, 0]
    out = torch.zeros((B, 2, 4), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    for kv in filter(None, args.fixed.split(",")):
        db.set_tuning(kv.split("=")[0], int(kv.split("=")[1]))
    name, vals = args.set.split("=")
    res = {"build": info, "table": db.table_layout(), "runs": []}
    print(json.dumps({"table": res["table"], "locations": int(db.info()[7])}), flush=True)
    ref = None
    nbt = args.nbatches
    outs = [out, torch.zeros_like(out)]
    for v, pipes in [(int(x), int(p)) for x in vals.split(",") for p in args.pipes.split(",")]:
        db.set_tuning(name, v)

        def step(i):
            r = db.query_device(batches[i % nbt].data_ptr(), qinfo.data_ptr(), B, B * bench.PAD_LEN, max_win_uniform=3)
            db.copy_results(out.data_ptr(), r.cands, B * 32)
            db.synchronize()

        pend = {}

        def finish2(j):
            if j in pend:
                db.query_finish(second_pipe=bool(j))
                db.copy_results(outs[j].data_ptr(), pend.pop(j), B * 32, second_pipe=bool(j))

        def step2(i):                                       # two batches in flight: enqueue batch i, THEN the tail of batch i - 1
            j = i & 1
            finish2(j)
            r = db.query_device(batches[i % nbt].data_ptr(), qinfo.data_ptr(), B, B * bench.PAD_LEN, max_win_uniform=3, second_pipe=bool(j), defer_tail=True)
            pend[j] = r.cands
            finish2(j ^ 1)

        run_step = step if pipes == 1 else step2
        run_step(0); run_step(1)
        finish2(0); finish2(1); db.synchronize()
        db.timing(True); db.timing_reset()
        t0 = time.perf_counter()
        for i in range(args.steps):
            run_step(i)
        finish2(0); finish2(1); db.synchronize()
        el = time.perf_counter() - t0
        db.timing(False)
        if pipes == 2:                                      # the same batch through the pipelined path: candidates must equal the sequential run's
            step2(0); finish2(0); finish2(1); db.synchronize()
        else:
            step(0)
        bs = db.last_batch_stats()
        c = out.clone()
        same = None if ref is None else bool(torch.equal(c, ref))
        if ref is None:
            ref = c
        kt = {k: db.timing_get(k) for k in bench.KERNELS}
        run = {name: v, "pipes": pipes, "ms_per_step": round(el / args.steps * 1e3, 3), "Mreads_per_min": round(B * args.steps / el * 60 / 1e6, 1),
               "same_candidates_as_first_setting": same, "stats": {k: bs[k] for k in ("locations", "filtered_kept", "filtered_reads", "filtered_over_512", "filter_second_kernel", "filter_handed_back")}, "kernel_ms": {k: round(x[0] / max(x[1], 1), 3) for k, x in kt.items() if x[0] > 0.02}}
        print(json.dumps(run), flush=True)
        res["runs"].append(run)
    if args.out:
        with open(args.out, "w") as f:
            json.dump(res, f, indent=1)
    db.close()


if __name__ == "__main__":
    main()
