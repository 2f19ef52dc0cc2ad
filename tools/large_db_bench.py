"""Scaled-down BASELINE configs[2] ("RefSeq-scale synthetic DB"): a phylogeny-shaped database far larger than the chip's
256 MB infinity cache -- species x strains (strains = mutated copies, so feature buckets are heavy-tailed up to the 254 cap),
uint32 target ids, built on this GPU by our builder -- queried with 150 bp reads resident in HBM.  Prints one JSON object with
the build/load times, table geometry, kernel times (HIP events) and the Mreads/min of the timed steps.  Not bench.py's `value`
(that is configs[1]); this is the experiment behind DESIGN.md §5 "large tables".

    python tools/large_db_bench.py [--species 500 --strains 4 --genome-len 5000000 --divergence 0.01]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from metacache_amd import api  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--species", type=int, default=500)
    ap.add_argument("--strains", type=int, default=4)
    ap.add_argument("--genome-len", type=int, default=5_000_000)
    ap.add_argument("--divergence", type=float, default=0.01)
    ap.add_argument("--batch", type=int, default=1_000_000)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--maxcand", type=int, default=2)
    ap.add_argument("--load-factor", type=float, default=0.3)
    ap.add_argument("--lowest", type=int, default=0, help="taxon rank for candidate merging (0 = sequence, 4 = species)")
    ap.add_argument("--passes", type=int, default=1, help="build the table in this many key shards (mc_build_finish_shards): beyond 2^32 pairs")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    G, GL = args.species * args.strains, args.genome_len
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(3100)
    gcat = torch.empty(G * GL, dtype=torch.uint8, device=dev)                      # all targets, ASCII, resident for read synthesis
    t0 = time.time()
    P = max(1, args.passes)
    blds = [api.Builder(target_id_bytes=4, max_candidates=args.maxcand, max_load_factor=args.load_factor,
                        **({"key_shard_index": 0, "key_shard_count": P} if P > 1 else {}))]
    t_add = 0.0
    for sp in range(args.species):
        base = torch.randint(0, 4, (GL,), generator=gen, device=dev, dtype=torch.uint8)
        for st in range(args.strains):
            code = base
            if st:
                sub = torch.rand(GL, generator=gen, device=dev) < args.divergence
                shift = torch.randint(1, 4, (GL,), generator=gen, device=dev, dtype=torch.uint8)
                code = torch.where(sub, (base + shift) % 4, base)
            t = sp * args.strains + st
            gcat[t * GL:(t + 1) * GL] = lut[code.long()]
            host = gcat[t * GL:(t + 1) * GL].cpu().numpy()
            t1 = time.time()
            blds[0].add_target(host, f"SYN_{t:06d}.1", parent_taxid=1000 + sp, filename=f"syn{t}.fa")
            t_add += time.time() - t1
    t_gen = time.time() - t0
    t1 = time.time()
    blds[0].finish(load=False)
    for p in range(1, P):                                      # the other key shards: the same targets again, from HBM
        b = api.Builder(target_id_bytes=4, max_candidates=args.maxcand, max_load_factor=args.load_factor, key_shard_index=p, key_shard_count=P)
        for t in range(G):
            b.add_target(gcat[t * GL:(t + 1) * GL].cpu().numpy(), f"SYN_{t:06d}.1", parent_taxid=1000 + t // args.strains, filename=f"syn{t}.fa")
        b.finish(load=False)
        blds.append(b)
    # reads are drawn before the genomes leave HBM (the table needs the room)
    B = args.batch
    goff = torch.arange(G, device=dev, dtype=torch.int64) * GL
    batches = [torch.cat([bench.synth_reads_gpu(gcat, goff, GL, B, seed=3100 + s).reshape(-1), torch.zeros(16, dtype=torch.uint8, device=dev)])
               for s in range(min(args.steps, 4))]
    if P > 1:
        del gcat
        torch.cuda.empty_cache()
    db = api.Builder.finish_shards(blds) if P > 1 else blds[0].finish(load=True)
    t_finish = time.time() - t1
    for b in blds:
        b.free()
    # taxonomy for merging above sequence level: species = parent
    if args.lowest:
        lin = np.zeros((G, 21), dtype=np.uint32)
        lin[:, 0] = np.arange(G) + 1                                                # every target its own sequence-level taxon
        lin[:, 4] = G + 1 + np.arange(G) // args.strains                            # species = index of the strain group
        db.set_lineages(lin)
    info = db.info()
    res = {"targets": G, "bases": G * GL, "strains_per_species": args.strains, "divergence": args.divergence,
           "db_info": {"k": info[0], "s": info[1], "w": info[2], "stride": info[3], "max_locs": info[4], "targets": info[5], "locations": info[7]},
           "build_passes": P, "seconds": {"generate_and_add_targets": round(t_gen, 2), "of_which_add_target": round(t_add, 2), "sort_rle_table": round(t_finish, 2)},
           "hbm_allocated_GB": round(torch.cuda.mem_get_info()[1] / 1e9 - torch.cuda.mem_get_info()[0] / 1e9, 2)}

    K = args.maxcand
    qinfo = torch.zeros((B, 4), dtype=torch.int32, device=dev)
    qinfo[:, 0] = torch.arange(B, device=dev, dtype=torch.int32) * bench.PAD_LEN
    qinfo[:, 1] = bench.READ_LEN
    qinfo[:, 2] = qinfo[:, 0]
    max_win = db.max_windows_in_range(bench.READ_LEN)
    out = torch.zeros((B, K, 4), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()

    qst = torch.zeros((B, 4), dtype=torch.int32, device=dev)

    def step(i):
        r = db.query_device(batches[i % len(batches)].data_ptr(), qinfo.data_ptr(), B, B * bench.PAD_LEN, max_win_uniform=max_win, lowest=args.lowest)
        db.copy_results(out.data_ptr(), r.cands, B * K * 16)
        db.copy_results(qst.data_ptr(), r.hit_counts, B * 16)
        db.synchronize()

    for i in range(2):
        step(i)
    db.timing(True); db.timing_reset()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    el = time.perf_counter() - t0
    db.timing(False)
    kt = {k: db.timing_get(k) for k in ("plan", "sketch_probe", "sketch_lane", "chunk_sketch", "chunk_probe", "probe_cands", "mid_cands_64", "mid_cands_128", "mid_cands_256", "hash_cands_256", "hash_cands_512", "hash_cands_1024", "query_wave", "scan", "sort_candidates")}
    st = db.last_batch_stats()
    c = out.cpu().numpy().view(np.uint32).reshape(B, K, 4)
    res["query"] = {"reads_per_step": B, "steps": args.steps, "ms_per_step": round(el / args.steps * 1e3, 3),
                    "Mreads_per_min": round(B * args.steps / el * 60 / 1e6, 1),
                    "kernel_ms": {k: round(v[0] / max(v[1], 1), 4) for k, v in kt.items()},
                    "features_per_read": round(st["features"] / B, 2), "locations_per_read": round(st["locations"] / B, 2),
                    "list_length_classes": {k: int(v) for k, v in zip(("<=32", "33-64", "65-128", "129-256", ">256"),
                                                                       np.histogram(qst[:, 0].cpu().numpy(), bins=[0, 33, 65, 129, 257, 1 << 30])[0])},
                    "stats": st, "reads_with_candidate": float((c[:, 0, 1] > 0).mean()),
                    "mean_top_hits": float(c[:, 0, 1].mean())}
    print(json.dumps(res, indent=1))
    if args.out:
        os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(res, f, indent=1)
    db.close()


if __name__ == "__main__":
    main()
