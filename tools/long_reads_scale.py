"""Long single reads (BASELINE configs[4]'s shape) against the configs[2] collection at --scale: time per batch through the C ABI and
parity of a sample against the oracle's restricted build.  python tools/long_reads_scale.py --scale 0.3 --lengths 500,2000,10000"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
from metacache_amd import synthdb  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=0.3)
    ap.add_argument("--lengths", default="500,2000,10000")
    ap.add_argument("--bases", type=int, default=200_000_000, help="bases per batch")
    ap.add_argument("--check", type=int, default=150)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    import scale_util
    dev = torch.device("cuda", 0)
    c2 = dict(bench.CFG2); c2["genera"] = max(2, int(round(c2["genera"] * args.scale)))
    spec = synthdb.phylogeny(**c2)
    shards = max(1, int(np.ceil(spec.total_bases // 112 * 16 / 1.4e9)))
    K = 2
    db, info = synthdb.build_database(spec, shards=shards, max_candidates=K, max_load_factor=0.3, report=lambda m: print(m, file=sys.stderr, flush=True))
    gen = synthdb.GpuSynth(0)
    res = {"scale": args.scale, "bases": int(spec.total_bases), "locations": int(db.info()[7]), "table": db.table_layout(), "runs": []}
    for L in [int(x) for x in args.lengths.split(",")]:
        n = max(args.check, args.bases // L)
        P = synthdb.read_params(spec, 5100 + L, read_len=L)
        rows = torch.zeros((n, P.row_bytes), dtype=torch.uint8, device=dev)
        gen.reads(spec, P, 0, n, rows)
        seq = torch.cat([rows.reshape(-1), torch.zeros(16, dtype=torch.uint8, device=dev)])
        qinfo = torch.zeros((n, 4), dtype=torch.int32, device=dev)
        qinfo[:, 0] = torch.arange(n, device=dev, dtype=torch.int32) * P.row_bytes
        qinfo[:, 1] = L; qinfo[:, 2] = qinfo[:, 0]
        mw = db.max_windows_in_range(L, 0)
        out = torch.zeros((n, K, 4), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()

        def step():
            r = db.query_device(seq.data_ptr(), qinfo.data_ptr(), n, n * P.row_bytes, max_win_uniform=mw)
            db.copy_results(out.data_ptr(), r.cands, n * K * 16)
            db.synchronize()
        step()
        db.timing(True); db.timing_reset()
        t0 = time.perf_counter()
        step(); step()
        el = (time.perf_counter() - t0) / 2
        db.timing(False)
        st = db.last_batch_stats()
        kt = {k: db.timing_get(k) for k in bench.KERNELS}
        host = rows[:args.check].cpu().numpy()
        reads = [bytes(r[:L]) for r in host]
        odb = scale_util.oracle_database(spec, scale_util.sample_features(reads), threads=max(4, 2 * scale_util.effective_cpus()))
        got = out[:args.check].cpu().numpy().view(np.uint32)
        bad = 0
        for i, r in enumerate(reads):
            _, e = odb.query(r, b"", K, 0, 0)
            e = e[:K]
            for k in range(K):
                g = tuple(int(x) for x in got[i, k]) if got[i, k, 1] else None
                x = (int(e[k]["tgt"]), int(e[k]["hits"]), int(e[k]["beg"]), int(e[k]["end"])) if k < len(e) else None
                bad += g != x
        odb.close()
        run = {"read_len": L, "reads": n, "max_windows_in_range": mw, "ms_per_batch": round(el * 1e3, 2), "Gbases_per_s": round(n * L / el / 1e9, 3),
               "reads_per_s": round(n / el), "locations_per_read": round(st["locations"] / n, 1), "checked": args.check, "mismatches": bad,
               "filtered": {k: st[k] for k in ("filtered_kept", "filtered_reads", "filtered_over_512", "filter_second_kernel", "filter_handed_back")},
               "kernel_ms": {k: round(v[0] / max(v[1], 1), 3) for k, v in kt.items() if v[0] > 0.05}}
        print(json.dumps(run), flush=True)
        res["runs"].append(run)
    if args.out:
        with open(args.out, "w") as f:
            json.dump(res, f, indent=1)
    db.close()


if __name__ == "__main__":
    main()
