"""The REFERENCE itself as the checker on a mid-size cut of the configs[2] collection: the database is built on the GPU in key shards,
written as database files (mc_build_write_shards), loaded by the reference (oracle/_ref, compiled from its sources) and queried on the
host threads; every candidate of every read is compared with the GPU's, and the reference's throughput is recorded for a thread sweep.
(At full scale the reference cannot do this: a 190 GB file set, loaded single-threaded.)  TEST / MEASUREMENT TOOL.

    python tools/ref_parity_midscale.py --scale 0.05 --reads 300000 --out gpurun_out/ref_parity_midscale.json
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import cpuref  # noqa: E402
import scale_util  # noqa: E402
from metacache_amd import api, synthdb  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=0.05)
    ap.add_argument("--reads", type=int, default=300_000)
    ap.add_argument("--shards", type=int, default=2)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    c2 = dict(bench.CFG2); c2["genera"] = max(2, int(round(c2["genera"] * args.scale)))
    spec = synthdb.phylogeny(**c2)
    n = len(spec.targets)
    gen = synthdb.GpuSynth(0)
    K = 2
    # all shards' builders stay alive: the files are written from them
    t0 = time.time()
    lens = spec.targets["length"].astype(np.int64)
    groups, first, acc = [], 0, 0
    for t in range(n):
        ln = (int(lens[t]) + 3) // 4 * 4
        if acc + ln > (2 << 30) and t > first:
            groups.append((first, t - first)); first, acc = t, 0
        acc += ln
    groups.append((first, n - first))
    buf = torch.zeros(max(int(spec.offsets(f, c)[-1]) for f, c in groups) + 64, dtype=torch.uint8, device=dev)
    builders = []
    for sh in range(args.shards):
        b = api.Builder(target_id_bytes=4, max_candidates=K, max_load_factor=0.5, key_shard_index=sh, key_shard_count=args.shards)
        for f, c in groups:
            off = gen.targets(spec, f, c, buf)
            for i in range(c):
                b.add_target_device(buf.data_ptr() + int(off[i]), int(lens[f + i]), f"SYN_{f + i:06d}.1", 1000 + int(spec.species[f + i]))
            b.flush()
        b.finish(load=False)
        builders.append(b)
    db = api.Builder.finish_shards(builders)
    t_build = time.time() - t0
    tmp = tempfile.mkdtemp(prefix="mcmid", dir="/tmp")
    name = os.path.join(tmp, "mid")
    t0 = time.time()
    api.Builder.write_shards(builders, name, spec.taxa())
    t_write = time.time() - t0
    for b in builders:
        b.free()
    del buf
    size = sum(os.path.getsize(name + e) for e in (".meta", ".cache0"))
    # reads + GPU candidates
    P = synthdb.read_params(spec, 3100)
    B = args.reads
    reads = torch.zeros(B * bench.PAD_LEN + 16, dtype=torch.uint8, device=dev)
    gen.reads(spec, P, 0, B, reads)
    qinfo = torch.zeros((B, 4), dtype=torch.int32, device=dev)
    qinfo[:, 0] = torch.arange(B, device=dev, dtype=torch.int32) * bench.PAD_LEN
    qinfo[:, 1] = bench.READ_LEN; qinfo[:, 2] = qinfo[:, 0]
    out = torch.zeros((B, K, 4), dtype=torch.int32, device=dev)
    r = db.query_device(reads.data_ptr(), qinfo.data_ptr(), B, B * bench.PAD_LEN, max_win_uniform=3)
    db.copy_results(out.data_ptr(), r.cands, B * K * 16); db.synchronize()
    st = db.last_batch_stats()
    g = out.cpu().numpy().view(np.uint32)
    gc = np.zeros((B, K), dtype=api.cand_dtype)
    gc["tgt"], gc["hits"], gc["beg"], gc["end"] = g[..., 0], g[..., 1], g[..., 2], g[..., 3]
    info = db.info()
    db.close()
    host = reads[: B * bench.PAD_LEN].reshape(B, bench.PAD_LEN).cpu().numpy()
    seqs = np.ascontiguousarray(host[:, :bench.READ_LEN]).reshape(-1)
    offs = np.arange(B + 1, dtype=np.uint64) * np.uint64(bench.READ_LEN)
    # the reference
    t0 = time.time()
    rdb = cpuref.reference(4).open(name)
    t_load = time.time() - t0
    eff = scale_util.effective_cpus()
    sweep = {}
    cands = None
    for th in [t for t in (1, 8, 16, 32) if t <= 2 * eff]:
        m = B if th >= 8 else min(B, 20000)
        el, c = rdb.query_many(seqs[: m * bench.READ_LEN], offs[: m + 1], max_cand=K, threads=th)
        sweep[th] = round(m / el * 60 / 1e6, 2)
        if m == B:
            cands = c
    rdb.close()
    mism = bench.count_mismatches(gc, cands)
    res = {"collection": {"targets": n, "bases": spec.total_bases, "locations": info[7], "database_file_bytes": size},
           "seconds": {"gpu_build": round(t_build, 1), "write_files": round(t_write, 1), "reference_load": round(t_load, 1)},
           "locations_per_read": round(st["locations"] / B, 1), "host_cpus_granted": eff,
           "reference_Mreads_per_min_by_threads": sweep,
           "parity": {"checked": B, "mismatches": mism, "against": "reference (oracle/_ref, database files written by this repo)"}}
    print(json.dumps(res, indent=1))
    if args.out:
        with open(args.out, "w") as f:
            json.dump(res, f, indent=1)
    for e in (".meta", ".cache0"):
        os.remove(name + e)
    sys.exit(0 if mism == 0 else 1)


if __name__ == "__main__":
    main()
