#!/usr/bin/env python3
"""bench.py -- MetaCache query hot path on MI355X: Mreads/min (150 bp) + roofline + CPU baseline.

Workload (BASELINE.json configs[1]): 16-genome synthetic DB (16 x 5 Mbp i.i.d. ACGT, seed 16, uint16
target ids, one partition), 10 M synthetic 150 bp reads (seed 1016; 1 % substitutions, 0.1 % N, both
strands).  A "step" = one pass of the hot path (sketch+probe -> scan -> sort+candidates) over one
batch of reads that is already resident in HBM; default 10 steps x 1 M reads = the 10 M reads.

N > 1 (launched by torch.distributed.run, one rank per GPU): the database is replicated, every rank
processes its own shard of reads (weak scaling: reads per GPU fixed) and the per-rank top-candidate
lists are gathered to rank 0 over RCCL inside the timed region -- the hand-over to host-side
taxonomy assignment.  No other collective: the path has no exchange step in this mode.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402  (plumbing: device memory, streams, torch.distributed)
import torch.distributed as dist  # noqa: E402

from metacache_amd import api, synth  # noqa: E402
from metacache_amd.distributed import gather_candidates_async  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak (MI355X_MICROARCH.md)
READ_LEN = 150
DEFAULT_BATCH = 10_000_000       # reads per step: the 10 M reads of configs[1] as one batch (4 M: 6 % slower per read, fixed per-batch costs)
PAD_LEN = 152                  # every read starts 4-byte aligned


def make_genomes(n_genomes: int, length: int, seed: int):
    rng = np.random.default_rng(seed)
    return [synth.random_genome(rng, length) for _ in range(n_genomes)]


def synth_reads_gpu(gcat: torch.Tensor, goff: torch.Tensor, glen: int, n: int, seed: int) -> torch.Tensor:
    """n reads of READ_LEN on the GPU: uniform genome, uniform start, random strand, 1 % substitutions,
    0.1 % N.  Returns uint8 [n, PAD_LEN] (zero padded)."""
    dev = gcat.device
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    gi = torch.randint(0, goff.numel(), (n,), generator=g, device=dev)
    st = torch.randint(0, glen - READ_LEN + 1, (n,), generator=g, device=dev)
    idx = (goff[gi] + st)[:, None] + torch.arange(READ_LEN, device=dev)[None, :]
    reads = gcat[idx]                                                  # [n, L] uint8 ASCII
    code = torch.zeros_like(reads)                                     # A0 C1 G2 T3
    code[reads == ord("C")] = 1; code[reads == ord("G")] = 2; code[reads == ord("T")] = 3
    flip = torch.rand(n, generator=g, device=dev) < 0.5
    rc = (3 - code).flip(1)
    code = torch.where(flip[:, None], rc, code)
    sub = torch.rand(code.shape, generator=g, device=dev) < 0.01
    shift = torch.randint(1, 4, code.shape, generator=g, device=dev, dtype=torch.uint8)
    code = torch.where(sub, (code + shift) % 4, code)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    out = lut[code.long()]
    nmask = torch.rand(code.shape, generator=g, device=dev) < 0.001
    out[nmask] = ord("N")
    padded = torch.zeros((n, PAD_LEN), dtype=torch.uint8, device=dev)
    padded[:, :READ_LEN] = out
    return padded


def measured_traffic(kernel_timer_name: str):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC summary of this
    workload (profiles/*_pmc_summary.csv: FETCH_SIZE / WRITE_SIZE in KB from separate --pmc passes).
    gfx950 caveat of MI355X_MICROARCH.md (FETCH_SIZE tallies every request at 64 B) calibrated for this
    kernel's access pattern in profiles/r01_fetch_calibration.md: probe_cands reads 64-byte buckets = 64-byte
    sector requests, which are counted at their true size, so no doubling here.  None if no summary exists."""
    import csv, glob
    names = {"sketch_lane": ("sketch_lane",), "probe_cands": ("probe_cands",), "sketch_probe": ("sketch_probe_lane",),
             "query_wave": ("query_kernel<fused>", "query_kernel<unfused>"), "sort_candidates": ("sort_candidates",)}.get(kernel_timer_name, ())
    for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.csv")), reverse=True):
        vals = {}
        for r in csv.DictReader(open(fn)):
            if r["kernel"] in names and r["counter"] in ("FETCH_SIZE", "WRITE_SIZE"):
                vals.setdefault(r["kernel"], {})[r["counter"]] = float(r["mean_per_dispatch"])
        for k in names:
            if k in vals and len(vals[k]) == 2:
                return (vals[k]["FETCH_SIZE"] + vals[k]["WRITE_SIZE"]) * 1024.0, os.path.basename(fn)
    return None, None


def algorithmic_bytes_per_read(F: float, H: float, K: int, V: int) -> float:
    """ceil(L/4) + ceil(L/8) + 12 F + V H + 16 K"""
    return (READ_LEN + 3) // 4 + (READ_LEN + 7) // 8 + 12.0 * F + V * H + 16.0 * K


def cpu_baseline(dbname: str, reads_host: np.ndarray, gpu_cands: np.ndarray, K: int, target_bytes: int, budget_s: float):
    """Times the reference (oracle/_ref, kind 'reference') or our C restatement (kind 'port') on this
    box's host cores on a bounded sample of the same reads, and compares its candidates with the GPU's."""
    import cpuref
    n_total = reads_host.shape[0]
    kind = "reference" if cpuref.have_reference(target_bytes) else "port"
    ref = cpuref.reference(target_bytes) if kind == "reference" else cpuref.oracle()
    cores = (os.cpu_count() or 1) if kind == "reference" else 1
    db = ref.open(dbname)

    def run(n):
        seqs = np.ascontiguousarray(reads_host[:n, :READ_LEN]).reshape(-1)
        offs = np.arange(n + 1, dtype=np.uint64) * np.uint64(READ_LEN)
        return db.query_many(seqs, offs, max_cand=K, lowest=0, insert_max=0, threads=cores)

    probe_n = min(n_total, 20000 * cores)
    t, _ = run(probe_n)
    rate = probe_n / max(t, 1e-9)
    n = int(min(n_total, max(probe_n, rate * budget_s)))
    t, cands = run(n)
    db.close()
    mism = 0
    g = gpu_cands[:n]
    for f in ("tgt", "hits", "beg", "end"):
        mism_f = (g[f] != cands[f]) & ((g["hits"] > 0) | (cands["hits"] > 0))
        mism = max(mism, int(mism_f.any(axis=1).sum()))
    return {"value": n / t * 60.0 / 1e6, "unit": "Mreads/min", "cores": cores, "kind": kind,
            "sample": f"{n} reads of the same workload (batch 0), {cores} host thread(s), database files written by this repo"}, \
           {"checked": n, "mismatches": mism, "against": kind}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=DEFAULT_BATCH, help="reads per step per GPU")
    ap.add_argument("--genomes", type=int, default=16)
    ap.add_argument("--genome-len", type=int, default=5_000_000)
    ap.add_argument("--maxcand", type=int, default=2)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU baseline leg (0 = skip)")
    ap.add_argument("--load-factor", type=float, default=0.3)
    ap.add_argument("--force-dist", action="store_true", help="run the N>1 gather path with a single rank too (testing)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1 or args.force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    K = args.maxcand
    B = args.batch
    # ---- database (replicated on every rank): built on the GPU by our builder ------------------
    genomes = make_genomes(args.genomes, args.genome_len, seed=16)
    t0 = time.time()
    bld = api.Builder(device=local, target_id_bytes=2, max_candidates=K, max_load_factor=args.load_factor)
    for i, gnm in enumerate(genomes):
        bld.add_target(gnm, f"SYN_{i:06d}.1", parent_taxid=1000 + i, filename=f"syn{i}.fa")
    db = bld.finish(load=True)
    build_s = time.time() - t0
    dbdir = None
    if rank == 0 and args.cpu_seconds > 0:
        dbdir = tempfile.mkdtemp(prefix="mcbench")
        taxa = [(1, 1, 20, "root")] + [(1000 + i, 1, 4, f"synthetic species {i}") for i in range(args.genomes)]
        bld.write(os.path.join(dbdir, "syn16"), taxa)
    bld.free()
    db_info = db.info()

    # ---- reads: resident in HBM before the timed region ----------------------------------------
    gcat = torch.from_numpy(np.concatenate(genomes)).to(dev)
    goff = torch.arange(args.genomes, device=dev, dtype=torch.int64) * args.genome_len
    nb = max(1, min(max(args.steps, args.warmup), 8))                            # distinct batches resident in HBM, reused cyclically
    batches = []
    for s in range(nb):
        batches.append(synth_reads_gpu(gcat, goff, args.genome_len, B, seed=1016 + 7919 * rank + s).reshape(-1))
    del gcat
    qinfo = torch.zeros((B, 4), dtype=torch.int32, device=dev)
    qinfo[:, 0] = torch.arange(B, device=dev, dtype=torch.int32) * PAD_LEN
    qinfo[:, 1] = READ_LEN
    qinfo[:, 2] = qinfo[:, 0]
    slack = torch.zeros(16, dtype=torch.uint8, device=dev)
    batches = [torch.cat([b, slack]) for b in batches]
    max_win = db.max_windows_in_range(READ_LEN)              # = 3 for 150 bp
    # N > 1: the per-rank candidate lists go to rank 0 over RCCL (north_star: "per-rank partial hit lists gathered over RCCL/xGMI
    # before host-side taxonomy assignment").  Two buffers per rank: the gather of batch i runs while batch i+1 is computed.
    dist_path = world > 1 or args.force_dist
    nbuf = 2 if dist_path else 1
    out_bufs = [torch.zeros((B, K, 4), dtype=torch.int32, device=dev) for _ in range(nbuf)]
    out_cands = out_bufs[0]
    recv = [[torch.zeros((B, K, 4), dtype=torch.int32, device=dev) for _ in range(world)] for _ in range(nbuf)] if dist_path and rank == 0 else None
    works = [None] * nbuf
    torch.cuda.synchronize()

    def finish(j: int):
        if works[j] is not None:
            works[j].wait()
            torch.cuda.current_stream().synchronize()
            works[j] = None

    def step(i: int):
        b = batches[i % nb]
        j = i % nbuf
        finish(j)                                            # the gather that used this buffer two batches ago
        res = db.query_device(b.data_ptr(), qinfo.data_ptr(), B, B * PAD_LEN, max_win_uniform=max_win)
        db.copy_results(out_bufs[j].data_ptr(), res.cands, B * K * 16)
        db.synchronize()
        if dist_path:
            works[j] = gather_candidates_async(out_bufs[j], recv[j] if recv is not None else None, dst=0)
        return res

    def drain():
        for j in range(nbuf):
            finish(j)

    for i in range(args.warmup):
        step(i)
    drain()
    torch.cuda.synchronize()
    db.timing(True)
    db.timing_reset()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    F_sum = H_sum = 0
    for i in range(args.steps):
        step(i)
    drain()                                                  # every gather has arrived on rank 0
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    db.timing(False)
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        kt = {k: db.timing_get(k) for k in ("plan", "sketch_probe", "sketch_lane", "chunk_sketch", "chunk_probe", "probe_cands", "mid_cands_64", "mid_cands_128", "mid_cands_256", "hash_cands_256", "hash_cands_512", "hash_cands_1024", "query_wave", "scan", "sort_candidates")}
        st = db.last_batch_stats()                            # of the last timed batch
        F, H = st["features"] / B, st["locations"] / B
        V = 6                                                 # uint16 target ids: 6-byte locations in the file format
        bytes_per_read = algorithmic_bytes_per_read(F, H, K, V)
        dom = max(("sketch_probe", "sketch_lane", "probe_cands", "mid_cands_64", "mid_cands_128", "mid_cands_256", "hash_cands_256", "hash_cands_512", "hash_cands_1024", "query_wave", "sort_candidates"), key=lambda k: kt[k][0])
        dom_ms = kt[dom][0] / max(kt[dom][1], 1)
        achieved = bytes_per_read * B / (dom_ms * 1e-3) / 1e9
        traffic, traffic_src = measured_traffic(dom) if B == DEFAULT_BATCH else (None, None)   # the committed PMC passes ran the default batch
        total_reads = world * args.steps * B
        value = total_reads / elapsed * 60.0 / 1e6
        result = {
            "metric": "Mreads/min (150 bp)", "value": round(value, 2), "unit": "Mreads/min", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"configs[1]: {args.genomes}x{args.genome_len} bp synthetic DB (uint16 target ids, 1 partition), "
                                   f"{total_reads} synthetic 150 bp reads", "reads_per_step_per_gpu": B,
                       "maxcand": K, "k": db.k, "sketchlen": db.s, "winlen": db.w, "winstride": db.stride,
                       "db_locations": db_info[7], "db_build_s": round(build_s, 2), "load_factor": args.load_factor,
                       "parallelism": f"replicated DB x{world}, reads sharded, RCCL gather of top candidates"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": round(bytes_per_read * B),
                         "bytes_per_read": round(bytes_per_read, 1), "F": round(F, 3), "H": round(H, 3),
                         "kernel_ms": {k: round(v[0] / max(v[1], 1), 4) for k, v in kt.items()}},
        }
        if world == 1 and args.cpu_seconds > 0:
            res = step(0)
            drain()
            db.synchronize()
            gpu_c = out_cands.cpu().numpy().view(np.uint32).reshape(B, K, 4)
            gc = np.zeros((B, K), dtype=api.cand_dtype)
            gc["tgt"], gc["hits"], gc["beg"], gc["end"] = gpu_c[..., 0], gpu_c[..., 1], gpu_c[..., 2], gpu_c[..., 3]
            reads_host = batches[0][: B * PAD_LEN].reshape(B, PAD_LEN).cpu().numpy()
            cb, par = cpu_baseline(os.path.join(dbdir, "syn16"), reads_host, gc, K, 2, args.cpu_seconds)
            result["cpu_baseline"] = cb
            result["parity"] = par
    db.close()
    # the JSON line is the LAST thing on stdout: RCCL announces itself through C stdio ("Librccl path : ..."), which every rank
    # flushes here, before the barrier and the line, instead of at process exit after it
    import ctypes
    ctypes.CDLL(None).fflush(None)
    sys.stdout.flush()
    if dist.is_initialized():
        dist.barrier()
    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
