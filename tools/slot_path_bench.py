"""The host slot API (mc_batch_add / submit / wait / clear: what the reference's consumer threads call, database_query.hpp:185-252) at the
REFERENCE's batch size: 4 096 reads per batch (options.hpp:229-232: 8 192 windows), T threads with a slot each, on bench.py's collection.
Reads start in ordinary host memory (mc_batch_add_bulk copies them into the slot's pinned buffer), every batch is H2D + kernels + D2H
of its candidates.  Reported per (threads, batch size): Mreads/min, batches per second, and -- once -- that the candidates are the
device path's.  Never bench.py's `value`.

    python tools/slot_path_bench.py --scale 1 --threads 8,16,32 --batch 4096,65536 --out gpurun_out/r06_slot_path.json"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--reads", type=int, default=4_000_000)
    ap.add_argument("--threads", default="8,16,32")
    ap.add_argument("--batch", default="4096")
    ap.add_argument("--seconds", type=float, default=3.0, help="per configuration")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    import torch
    import bench
    from metacache_amd import api, synthdb
    c2 = dict(bench.CFG2); c2["genera"] = max(2, int(round(c2["genera"] * args.scale)))
    spec = synthdb.phylogeny(**c2)
    shards = max(1, int(np.ceil(spec.total_bases // 112 * 16 / 1.4e9)))
    threads = [int(t) for t in args.threads.split(",")]
    batches = [int(b) for b in args.batch.split(",")]
    P = synthdb.read_params(spec, 3100)
    rows = torch.zeros((args.reads, P.row_bytes), dtype=torch.uint8, device="cuda")
    synthdb.GpuSynth(0).reads(spec, P, 0, args.reads, rows)
    seqs = np.ascontiguousarray(rows[:, :150].cpu().numpy()).reshape(-1)
    del rows
    res = {"scale": args.scale, "Gbp": round(spec.total_bases / 1e9, 1), "reads_resident_on_the_host": args.reads, "runs": [],
           "dispatch": os.environ.get("AMD_DIRECT_DISPATCH", "default"), "coalesce": os.environ.get("MC_SLOT_COALESCE", "default")}
    L = api.lib()
    from metacache_amd import build
    drv = C.CDLL(build.build_slot_driver())
    drv.mc_slot_drive.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_double, C.c_void_p, C.c_uint64, C.c_void_p]
    for B in batches:
        # (a table per slot size: the slots' capacity decides whether the library unites them -- mc_slot_stats, include/metacache_amd.h)
        db, _ = synthdb.build_database(spec, shards=shards, max_candidates=2, max_load_factor=0.3, num_slots=max(threads), slot_max_queries=B,
                                       slot_max_chars=B * 152 + 64, report=lambda m: print(m, file=sys.stderr, flush=True))
        offs = np.arange(B + 1, dtype=np.uint64) * np.uint64(150)
        nb = args.reads // B
        # the device path's candidates of the first batches: what every slot must deliver
        want = db.query_bulk(seqs[: 150 * B * min(nb, 8)], np.arange(B * min(nb, 8) + 1, dtype=np.uint64) * np.uint64(150))
        wantc = np.ascontiguousarray(want)
        # (half a second untimed: the first enqueues of a process may fall into the runtime's slow submission state -- DESIGN 9 -- for a second or two)
        drv.mc_slot_drive(db.h, seqs.ctypes.data_as(C.c_void_p), args.reads, 150, B, max(threads), C.c_double(0.5), wantc.ctypes.data_as(C.c_void_p), len(wantc), (C.c_uint64 * 4)())
        for T in threads:
            o = (C.c_uint64 * 4)()
            rc = drv.mc_slot_drive(db.h, seqs.ctypes.data_as(C.c_void_p), args.reads, 150, B, T, C.c_double(args.seconds), wantc.ctypes.data_as(C.c_void_p), len(wantc), o)
            done, bad, errs, el = [int(o[0])], [int(o[1])], ([f"rc {rc}, {int(o[2])} threads failed: " + L.mc_last_error(db.h).decode()] if rc else []), o[3] / 1e6
            n = sum(done) * B
            st = (C.c_uint64 * 4)()
            L.mc_slot_stats.argtypes = [C.c_void_p, C.c_void_p]
            L.mc_slot_stats(db.h, st)
            run = {"slots_united": bool(st[0]), "united_batches_so_far": int(st[1]), "slots_carried_so_far": int(st[2]), "threads": T, "batch": B, "batches": sum(done), "seconds": round(el, 2), "Mreads_min": round(n / el * 60 / 1e6, 1),
                   "batches_per_s": round(sum(done) / el), "us_per_batch_and_thread": round(el / max(1, sum(done)) * T * 1e6), "reads_with_other_candidates": sum(bad), "errors": errs[:2]}
            print(run, flush=True)
            res["runs"].append(run)
        db.close()
    print(json.dumps(res))
    if args.out:
        os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
        json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
