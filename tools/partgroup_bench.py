"""Part groups where loading matters: bench.py's collection at --scale cut into --parts parts (targets dealt out round-robin, as the
reference's -parts), every part written as its own .cache file of ONE partitioned database in /dev/shm, then a read set classified
  (a) with --resident parts in HBM at a time (mc_partset_*: the next group loads behind this group's queries), and
  (b) with all parts resident,
candidates compared; reported: bytes loaded, seconds and GB/s of the group loads, seconds the queries waited for the loader, ms per batch.
Reference: docs/partitioning.md:116-153 (query part by part, merge), database.cpp:203-226 (one thread per part).
  python tools/partgroup_bench.py --scale 0.8 --parts 8 --resident 2 --reads 1000000 --out profiles/r04_partgroups.json"""
import argparse
import json
import os
import struct
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=0.8)
    ap.add_argument("--parts", type=int, default=8)
    ap.add_argument("--resident", type=int, default=2)
    ap.add_argument("--reads", type=int, default=1_000_000)
    ap.add_argument("--batch", type=int, default=250_000)
    ap.add_argument("--env-variants", default="", help="'A=1,B=2;A=3': the part-group run once more per combination of environment switches (MC_LOAD_THREADS, MC_PARTSET_LOADS_PER_DEVICE)")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    import torch
    from metacache_amd import api, synthdb
    import bench
    shm = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
    name = os.path.join(shm, f"mcparts_{os.getpid()}")
    c2 = dict(bench.CFG2); c2["genera"] = max(2, int(round(c2["genera"] * args.scale)))
    spec = synthdb.phylogeny(**c2)
    N = args.parts
    res = {"scale": args.scale, "Gbp": round(spec.total_bases / 1e9, 2), "parts": N, "resident": args.resident, "reads": args.reads}
    made = []
    try:
        t0 = time.time()
        for p in range(N):
            sel = np.arange(p, len(spec.targets), N)
            part_pairs = int(spec.targets["length"][sel].sum()) // 112 * 16
            shards = max(1, int(np.ceil(part_pairs / 1.4e9)))
            tmp = f"{name}_p{p}"
            db, _ = synthdb.build_database(spec, shards=shards, max_candidates=2, max_load_factor=0.3, write_to=tmp, only_targets=sel)
            db.close()
            os.replace(tmp + ".cache0", f"{name}.cache{p}"); made.append(f"{name}.cache{p}")
            if p == 0:
                os.replace(tmp + ".meta", name + ".meta"); made.append(name + ".meta")
                # number of parts in the metadata (database.cpp:247-290: version u64 | 7 type widths | sketching 2 x 4 u64 | max locations u64 |
                # target count (u32 here) | parts u32)
                with open(name + ".meta", "r+b") as f:
                    f.seek(8 + 7 + 64 + 8 + 4)
                    assert struct.unpack("<I", f.read(4))[0] == 1
                    f.seek(8 + 7 + 64 + 8 + 4)
                    f.write(struct.pack("<I", N))
            else:
                os.remove(tmp + ".meta")
            print(f"part {p + 1}/{N} written, {time.time() - t0:.0f} s", file=sys.stderr, flush=True)
        sizes = [os.path.getsize(f"{name}.cache{p}") for p in range(N)]
        res["part_file_GB"] = [round(s / 1e9, 2) for s in sizes]
        res["build_and_write_s"] = round(time.time() - t0, 1)
        # one plain read of the files first: the FIRST read of tmpfs pages that were just written is slower than every later one (11 - 15 GB/s
        # against 25 - 41 with the same eight threads, MC_LOAD_TRACE) -- the runs below then see files as a page cache holds them
        from concurrent.futures import ThreadPoolExecutor

        def slurp(fn):
            n = 0
            with open(fn, "rb", buffering=0) as f:
                buf = bytearray(64 << 20)
                while True:
                    k = f.readinto(buf)
                    if not k:
                        return n
                    n += k
        tw = time.time()
        with ThreadPoolExecutor(max_workers=8) as ex:
            total = sum(ex.map(slurp, [f"{name}.cache{p}" for p in range(N)]))
        res["first_read_after_write"] = {"GB": round(total / 1e9, 2), "seconds": round(time.time() - tw, 2), "GB_per_s": round(total / 1e9 / (time.time() - tw), 2), "threads": 8}
        # the reads (host memory: mc_partset_classify_resident takes host buffers, as the command line does)
        P = synthdb.read_params(spec, 3100)
        rows = torch.zeros((args.reads, P.row_bytes), dtype=torch.uint8, device="cuda")
        synthdb.GpuSynth(0).reads(spec, P, 0, args.reads, rows)
        host = rows.cpu().numpy()
        del rows
        torch.cuda.empty_cache()
        reads = [bytes(host[i, :150]) for i in range(args.reads)]
        K = 2

        def run(resident):
            t_open = time.time()
            ps = api.PartSet(name, resident=resident, devices=[0], max_candidates=K, slot_max_queries=args.batch, slot_max_chars=args.batch * 160)
            open_s = time.time() - t_open
            groups = ps.info()["groups"]
            out = np.zeros((args.reads, K), dtype=api.cand_dtype)
            per_group = []
            t_all = time.time()
            for g in range(groups):
                t0 = time.time()
                ps.select_group(g)
                t1 = time.time()
                for lo in range(0, args.reads, args.batch):
                    hi = min(args.reads, lo + args.batch)
                    ps.classify_resident(reads[lo:hi], None, out[lo:hi], has_prior=g > 0)
                t2 = time.time()
                per_group.append({"select_s": round(t1 - t0, 3), "query_s": round(t2 - t1, 3)})
            total = time.time() - t_all
            info = ps.info()
            ps.close()
            nb = -(-args.reads // args.batch)
            return out, {"resident": resident, "groups": groups, "open_first_group_s": round(open_s, 3), "all_groups_s": round(total, 3), "per_group": per_group,
                         "loader_s": round(info["load_s"], 3), "waited_for_loader_s": round(info["wait_s"], 3), "loaded_GB": round(info["load_bytes"] / 1e9, 2),
                         "load_GB_per_s": round(info["load_bytes"] / 1e9 / max(info["load_s"], 1e-9), 2),
                         "ms_per_batch_per_group": round(sum(g["query_s"] for g in per_group) / (groups * nb) * 1e3, 2)}
        variants = []
        for combo in filter(None, args.env_variants.split(";")):
            kv = dict(x.split("=") for x in combo.split(","))
            os.environ.update(kv)
            _, rv = run(args.resident)
            rv["env"] = kv
            variants.append(rv)
            print(json.dumps(rv), flush=True)
            for k in kv:
                os.environ.pop(k, None)
        if variants:
            res["part_groups_by_env"] = variants
        got, r1 = run(args.resident)
        print(json.dumps(r1), flush=True)
        ref, r2 = run(N)
        print(json.dumps(r2), flush=True)
        bad = 0
        for f in ("tgt", "hits", "beg", "end"):
            bad += int((((got[f] != ref[f]) & ((got["hits"] > 0) | (ref["hits"] > 0))).any(axis=1)).sum())
        res["part_groups"] = r1
        res["all_resident"] = r2
        res["reads_with_different_candidates"] = bad
    finally:
        for f in made:
            if os.path.exists(f):
                os.remove(f)
        for p in range(N):
            for e in (".meta", ".cache0"):
                if os.path.exists(f"{name}_p{p}{e}"):
                    os.remove(f"{name}_p{p}{e}")
    if args.out:
        with open(args.out, "w") as f:
            json.dump(res, f, indent=1)
    print(json.dumps({k: v for k, v in res.items() if k not in ("part_groups", "all_resident")}))


if __name__ == "__main__":
    main()
