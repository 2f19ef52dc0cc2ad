"""Database load speed: bench.py's collection at --scale written as database files to /dev/shm (GPU builder + streaming writer), then
mc_open_database timed with the pipelined loader (reader threads -> pinned slabs -> copy stream -> table kernels, dbload.cpp) for a few
thread counts and with the sequential loader of round 3 (MC_LOAD_PIPELINE=0), each in a fresh process; a read sample classified after
every load must give the same candidates.  Reference: database.cpp:203-226 (one thread per part), hash_multimap.hpp:970-1030.
  python tools/load_bench.py --scale 0.2 --out profiles/r04_load.json"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(name, n_reads):
    import numpy as np
    import torch
    from metacache_amd import api, synthdb
    import bench
    t0 = time.time()
    db = api.Database.open(name, max_candidates=2)
    open_s = time.time() - t0
    st = db.load_stats()
    c2 = dict(bench.CFG2); c2["genera"] = max(2, int(round(c2["genera"] * float(os.environ["MC_LB_SCALE"]))))
    spec = synthdb.phylogeny(**c2)
    P = synthdb.read_params(spec, 3100)
    rows = torch.zeros((n_reads, P.row_bytes), dtype=torch.uint8, device="cuda")
    synthdb.GpuSynth(0).reads(spec, P, 0, n_reads, rows)
    cands, _, _ = db.query([bytes(r[:150]) for r in rows.cpu().numpy()])
    import hashlib
    h = hashlib.sha256(np.ascontiguousarray(cands).tobytes()).hexdigest()[:16]
    print(json.dumps({"open_s": round(open_s, 3), "load": {k: (round(v, 4) if isinstance(v, float) else v) for k, v in st.items()}, "layout": db.table_layout(),
                      "candidates_sha": h}), flush=True)
    db.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=0.2)
    ap.add_argument("--threads", default="1,4,8,16")
    ap.add_argument("--reads", type=int, default=20000)
    ap.add_argument("--out", default="")
    ap.add_argument("--child", default="")
    args = ap.parse_args()
    if args.child:
        return child(args.child, args.reads)
    import numpy as np
    from metacache_amd import synthdb
    import bench
    shm = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
    name = os.path.join(shm, f"mcload_{os.getpid()}")
    c2 = dict(bench.CFG2); c2["genera"] = max(2, int(round(c2["genera"] * args.scale)))
    spec = synthdb.phylogeny(**c2)
    shards = max(1, int(np.ceil(spec.total_bases // 112 * 16 / 1.4e9)))
    t0 = time.time()
    db, info = synthdb.build_database(spec, shards=shards, max_candidates=2, max_load_factor=0.3, write_to=name,
                                      report=lambda m: print(m, file=sys.stderr, flush=True))
    db.close()
    size = sum(os.path.getsize(name + e) for e in (".meta", ".cache0"))
    res = {"scale": args.scale, "Gbp": round(spec.total_bases / 1e9, 2), "file_GB": round(size / 1e9, 2), "build_and_write_s": round(time.time() - t0, 1), "runs": []}
    try:
        runs = [("sequential (round 3)", {"MC_LOAD_PIPELINE": "0"})] + [(f"pipelined, {t} reader threads", {"MC_LOAD_THREADS": t}) for t in args.threads.split(",")]
        for label, env in runs:
            e = dict(os.environ, MC_LB_SCALE=str(args.scale), **env)
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", name, "--reads", str(args.reads)], env=e, capture_output=True, text=True)
            line = [l for l in p.stdout.split("\n") if l.startswith("{")]
            r = json.loads(line[-1]) if line else {"error": p.stderr[-500:]}
            r["loader"] = label
            if "open_s" in r:
                r["GB_per_s_open"] = round(size / 1e9 / r["open_s"], 2)
            print(json.dumps(r), flush=True)
            res["runs"].append(r)
        shas = {r.get("candidates_sha") for r in res["runs"]}
        res["same_candidates_after_every_load"] = len(shas) == 1 and None not in shas
    finally:
        for e in (".meta", ".cache0"):
            if os.path.exists(name + e):
                os.remove(name + e)
    if args.out:
        with open(args.out, "w") as f:
            json.dump(res, f, indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != "runs"}))


if __name__ == "__main__":
    main()
