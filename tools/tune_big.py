"""Builds the configs[2] table once (bench.py's collection at --scale) and times the hot path for several settings of the list-length
threshold of the filtered candidate path (big_min) -- the experiment behind the default.  python tools/tune_big.py --scale 1 --big-min 1024,512,256"""
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
    ap.add_argument("--batch", type=int, default=2_000_000)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--big-min", default="1024,512,256")
    ap.add_argument("--load-factor", type=float, default=0.5)
    ap.add_argument("--out", default="")
    ap.add_argument("--two-pipes", action="store_true", help="after the single-caller runs: two host threads, one batch in flight each (MC_SECOND_PIPE)")
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
    for s in range(2):
        t = torch.zeros(B * bench.PAD_LEN + 16, dtype=torch.uint8, device=dev)
        gen.reads(spec, P, s * B, B, t)
        batches.append(t)
    qinfo = torch.zeros((B, 4), dtype=torch.int32, device=dev)
    qinfo[:, 0] = torch.arange(B, device=dev, dtype=torch.int32) * bench.PAD_LEN
    qinfo[:, 1] = bench.READ_LEN; qinfo[:, 2] = qinfo[:, 0]
    out = torch.zeros((B, 2, 4), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    res = {"build": info, "runs": []}
    ref = None
    for bm in [int(x) for x in args.big_min.split(",")]:
        db.set_tuning("big_min", bm)

        def step(i):
            r = db.query_device(batches[i % 2].data_ptr(), qinfo.data_ptr(), B, B * bench.PAD_LEN, max_win_uniform=3)
            db.copy_results(out.data_ptr(), r.cands, B * 32)
            db.synchronize()
        step(0); step(1)
        db.timing(True); db.timing_reset()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i)
        el = time.perf_counter() - t0
        db.timing(False)
        step(0)
        c = out.clone()
        same = None if ref is None else bool(torch.equal(c, ref))
        if ref is None:
            ref = c
        kt = {k: db.timing_get(k) for k in bench.KERNELS}
        run = {"big_min": bm, "ms_per_step": round(el / args.steps * 1e3, 3), "Mreads_per_min": round(B * args.steps / el * 60 / 1e6, 1),
               "same_candidates_as_first_setting": same, "kernel_ms": {k: round(v[0] / max(v[1], 1), 3) for k, v in kt.items() if v[0] > 0.02}}
        print(json.dumps(run), flush=True)
        res["runs"].append(run)
    if args.two_pipes:
        import threading
        outs = [torch.zeros((B, 2, 4), dtype=torch.int32, device=dev) for _ in range(2)]
        streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
        n_steps = max(4, 2 * args.steps)

        def worker(t, n):
            st = streams[t]
            for i in range(t, n, 2):
                r = db.query_device(batches[i % 2].data_ptr(), qinfo.data_ptr(), B, B * bench.PAD_LEN, max_win_uniform=3, stream=st.cuda_stream,
                                    second_pipe=(t == 1))
                db.copy_results(outs[t].data_ptr(), r.cands, B * 32, stream=st.cuda_stream)
                st.synchronize()

        def run_two(n):
            th = [threading.Thread(target=worker, args=(t, n)) for t in range(2)]
            t0 = time.perf_counter()
            for x in th:
                x.start()
            for x in th:
                x.join()
            return time.perf_counter() - t0
        run_two(4)
        el = run_two(n_steps)
        same = bool(torch.equal(outs[0], ref)) and bool(torch.equal(outs[1][: B], (lambda: (step(1), out.clone())[1])()))
        run = {"two_pipes": True, "ms_per_step": round(el / n_steps * 1e3, 3), "Mreads_per_min": round(B * n_steps / el * 60 / 1e6, 1),
               "same_candidates": same}
        print(json.dumps(run), flush=True)
        res["runs"].append(run)
    if args.out:
        with open(args.out, "w") as f:
            json.dump(res, f, indent=1)
    db.close()


if __name__ == "__main__":
    main()
