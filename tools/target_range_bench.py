"""Mode T beside the whole table: bench.py's collection at --scale written as ONE database file, then a read set classified
  (a) by the whole table (one context), and
  (b) by --ranges contexts that each hold the locations of one contiguous target range of the SAME file (mc_config.target_shard_*:
      cut at load), all resident on the one GPU, per-range top lists gathered and merged in range order (mc_partset_*),
candidates compared read by read; reported: HBM per range (buckets, list store), seconds to load, ms per 10^6 reads, bytes per read a
rank hands to the gather (16 x max_candidates per range) beside what mode K's exchange moves for the same reads (4 bytes per location).
  python tools/target_range_bench.py --scale 0.1 --ranges 8 --reads 1000000 --out profiles/r05_target_ranges.json"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=0.1)
    ap.add_argument("--ranges", type=int, default=8)
    ap.add_argument("--reads", type=int, default=1_000_000)
    ap.add_argument("--batch", type=int, default=250_000)
    ap.add_argument("--resident", type=int, default=0, help="ranges in HBM at a time (0 = all; fewer: range groups, loaded one after the other)")
    ap.add_argument("--key-shards", type=int, default=0, help="also: the same file as this many key shards through mc_keyset_*")
    ap.add_argument("--skip-whole", action="store_true", help="no run of the whole table through the part set driver")
    ap.add_argument("--whole-only", action="store_true", help="only the whole table through the part set driver (e.g. with --reads 4000000: the driver's steady state)")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    import torch
    from metacache_amd import api, synthdb
    import bench
    shm = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
    name = os.path.join(shm, f"mcranges_{os.getpid()}")
    c2 = dict(bench.CFG2); c2["genera"] = max(2, int(round(c2["genera"] * args.scale)))
    spec = synthdb.phylogeny(**c2)
    K = 2
    res = {"scale": args.scale, "Gbp": round(spec.total_bases / 1e9, 2), "targets": len(spec.targets), "ranges": args.ranges, "reads": args.reads, "batch": args.batch}
    try:
        t0 = time.time()
        pairs = spec.total_bases // 112 * 16
        db, _ = synthdb.build_database(spec, shards=max(1, int(np.ceil(pairs / 1.4e9))), max_candidates=K, max_load_factor=0.3, write_to=name)
        db.close()
        res["file_GB"] = round(os.path.getsize(name + ".cache0") / 1e9, 2)
        res["build_and_write_s"] = round(time.time() - t0, 1)
        P = synthdb.read_params(spec, 3100)
        rows = torch.zeros((args.reads, P.row_bytes), dtype=torch.uint8, device="cuda")
        synthdb.GpuSynth(0).reads(spec, P, 0, args.reads, rows)
        seqs = np.ascontiguousarray(rows[:, :150].cpu().numpy()).reshape(-1)
        offs = np.arange(args.reads + 1, dtype=np.uint64) * 150
        # the same reads resident in HBM (bench.py's form: rows of 152 bytes), for the kernels' own time per context
        dev_rows = torch.cat([rows.reshape(-1), torch.zeros(16, dtype=torch.uint8, device="cuda")])
        del rows
        qinfo = torch.zeros((args.reads, 4), dtype=torch.int32, device="cuda")
        qinfo[:, 0] = torch.arange(args.reads, device="cuda", dtype=torch.int32) * P.row_bytes
        qinfo[:, 1] = 150
        qinfo[:, 2] = qinfo[:, 0]

        def device_ms(d):
            mw = d.max_windows_in_range(150, 0)
            best = 1e9
            for _ in range(4):
                torch.cuda.synchronize()
                t1 = time.time()
                d.query_device(dev_rows.data_ptr(), qinfo.data_ptr(), args.reads, args.reads * P.row_bytes, max_win_uniform=mw)
                d.synchronize()
                best = min(best, time.time() - t1)
            return round(best / args.reads * 1e9, 2)

        def hbm_used():
            free, total = torch.cuda.mem_get_info()
            return total - free

        def run(ranges):
            kw = dict(target_shard_count=ranges) if ranges > 1 else {}
            resident = max(1, min(ranges, args.resident or ranges))
            base = hbm_used()
            t_open = time.time()
            ps = api.PartSet(name, resident=resident, devices=[0], max_candidates=K, slot_max_queries=args.batch, slot_max_chars=args.batch * 160, **kw)
            open_s = time.time() - t_open
            used = hbm_used() - base
            groups = ps.info()["groups"]
            out = np.zeros((args.reads, K), dtype=api.cand_dtype)
            reads = [bytes(seqs[i * 150:(i + 1) * 150]) for i in range(min(args.reads, 10_000))]
            query_s, select_s, group_ms = [], [], []
            for rep in range(3 if groups == 1 else 1):
                q = 0.0
                for g in range(groups):
                    t0 = time.time()
                    ps.select_group(g)
                    select_s.append(round(time.time() - t0, 2))
                    if rep == 0:
                        ps.classify_resident(reads, None, np.zeros((len(reads), K), dtype=api.cand_dtype), has_prior=False)    # warm-up
                    t1 = time.time()
                    ps.classify_resident_packed(seqs, offs, out, has_prior=g > 0)      # (one call: --batch reads per batch, the batches pipelined)
                    q += time.time() - t1
                    group_ms.append(round((time.time() - t1) * 1e3, 1))     # (groups before the last: the next group loads behind these queries)
                query_s.append(q)
            info = ps.info()
            ps.close()
            return out, {"contexts": max(ranges, 1), "resident": resident, "groups": groups, "open_first_group_s": round(open_s, 2), "select_group_s": select_s,
                         "file_GB_read": round(info["load_bytes"] / 1e9, 2), "hbm_GB_resident_contexts_and_batch_buffers": round(used / 1e9, 2),
                         "ms_per_1e6_reads_all_ranges": round(min(query_s) / args.reads * 1e9, 2), "runs_ms": [round(x * 1e3, 1) for x in query_s], "per_group_ms": group_ms}

        ref, r1 = run(1) if not args.skip_whole else (None, {})
        print(json.dumps(r1), flush=True)
        if args.whole_only:
            res["whole_table"] = r1
            if args.out:
                os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
                json.dump(res, open(args.out, "w"), indent=1)
            return
        got, r2 = run(args.ranges)
        print(json.dumps(r2), flush=True)
        bad = 0
        for f in (("tgt", "hits", "beg", "end") if ref is not None else ()):
            bad += int((((got[f] != ref[f]) & ((got["hits"] > 0) | (ref["hits"] > 0))).any(axis=1)).sum())
        if args.key_shards:
            # mode K through ITS driver on the same file and reads (mc_keyset_*: the shards on the one GPU, the exchange as device copies)
            ks = api.KeySet(name, shards=args.key_shards, devices=[0], max_candidates=K, slot_max_queries=args.batch, slot_max_chars=args.batch * 160)
            ks.classify_packed(seqs[:150 * 10_000], offs[:10_001])
            tk = []
            for _ in range(3):
                t1 = time.time()
                gk = ks.classify_packed(seqs, offs)
                tk.append(time.time() - t1)
            ks.close()
            badk = 0
            for f in (("tgt", "hits", "beg", "end") if ref is not None else ()):
                badk += int((((gk[f] != ref[f]) & ((gk["hits"] > 0) | (ref["hits"] > 0))).any(axis=1)).sum())
            res["key_shards"] = {"shards": args.key_shards, "ms_per_1e6_reads_all_shards": round(min(tk) / args.reads * 1e9, 2), "runs_ms": [round(x * 1e3, 1) for x in tk],
                                 "reads_with_different_candidates": badk}
            print(json.dumps(res["key_shards"]), flush=True)
        res["whole_table"] = r1
        res["target_ranges"] = r2
        res["reads_with_different_candidates"] = bad
        # every range by itself: what it holds
        per = []
        whole = api.Database.open(name, max_candidates=K)
        lw = whole.table_layout()
        whole.target_range()
        _, counts, _ = whole.query([bytes(seqs[i * 150:(i + 1) * 150]) for i in range(20_000)])
        whole_ms = device_ms(whole)
        res["whole_table_layout"] = {"ms_per_1e6_reads_device_resident": whole_ms, "locations": whole.n_locations, "features": whole.n_features, "buckets": lw["buckets"], "list_store_entries": lw["list_locations"], "list_align": lw["list_align"],
                                     "GB": round((lw["buckets"] * 64 + lw["list_locations"] * lw["location_bytes"]) / 1e9, 2)}
        whole.close()
        for r in range(args.ranges):
            t1 = time.time()
            d = api.Database.open(name, max_candidates=K, target_shard_index=r, target_shard_count=args.ranges)
            l = d.table_layout()
            lo, hi = d.target_range()
            per.append({"ms_per_1e6_reads_device_resident": device_ms(d), "targets": [lo, hi], "locations": d.n_locations, "features": d.n_features, "load_factor": round(d.n_features / (4.0 * l["buckets"]), 3), "buckets": l["buckets"], "list_store_entries": l["list_locations"], "list_align": l["list_align"],
                        "GB": round((l["buckets"] * 64 + l["list_locations"] * l["location_bytes"]) / 1e9, 2), "open_s": round(time.time() - t1, 2)})
            d.close()
        res["per_range"] = per
        res["locations_in_ranges_over_whole"] = round(sum(p["locations"] for p in per) / max(res["whole_table_layout"]["locations"], 1), 6)
        loc = float(np.mean(counts))
        res["wire_bytes_per_read"] = {"mode_T_gather_per_rank": 16 * K, "mode_T_all_ranks_to_the_merging_rank": 16 * K * args.ranges,
                                      "mode_K_numbers_4_bytes_each": round(4 * loc, 1), "locations_per_read": round(loc, 1)}
    finally:
        for e in (".meta", ".cache0"):
            if os.path.exists(name + e):
                os.remove(name + e)
    if args.out:
        with open(args.out, "w") as f:
            json.dump(res, f, indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
