#!/usr/bin/env python3
"""bench.py -- MetaCache query hot path on MI355X: Mreads/min (150 bp) + roofline + CPU baseline.

Default workload = BASELINE.json configs[2]: a RefSeq-scale synthetic database (SURVEY.md §8d "Config 3": 40 000 targets,
150 Gbp, a genus -> species -> strain phylogeny with 0.5-10 % divergence, uint32 target ids, k = 16) built ON the GPU by this
repo's builder in key shards (the targets are a pure function of (target, position), metacache_amd/synth: generated in HBM group
by group, never resident as a whole), queried with synthetic 150 bp reads (1 % substitutions, 0.1 % N, both strands; seed 3100)
that are resident in HBM before the timed region.  A "step" = one pass of the hot path (sketch -> probe -> candidates) over one
batch of reads; the driver's `--steps 20` x 5 M reads = the 100 M reads of the configuration.
`--config 1` = configs[1] (16 x 5 Mbp, uint16 targets, 10 M reads per step) as in round 1; `--scale f` shrinks configs[2] for
quick runs (f = 1: 2000 genera; the line then says so in config.workload).

N > 1 (launched by torch.distributed.run, one rank per GPU), mode R: the database is replicated (every rank builds it), every
rank processes its own reads (weak scaling) and the per-rank top-candidate lists are gathered to rank 0 over RCCL inside the
timed region -- the hand-over to host-side taxonomy assignment.  No other collective: the path has no exchange step in this mode.

`--mode P` / `--mode K` run the two sharded forms of the path instead: P = the targets dealt out round-robin into N parts,
one part per rank, every rank classifies all reads of the step against its part, all-gather of the per-part top candidates, merge;
K = ONE part whose features are key-sharded over the ranks, every rank looks all reads up in its shard, all-to-all of the partial
location lists to the reads' owner ranks, union + candidates there (mc_candidates_from_partial_hits), gather to rank 0.
`--pairs` = configs[3]'s reads: 2 x 150 bp pairs (fragment 300-500, seed 4100, maxWindowsInRange 4); a pair counts as 2 reads.

Checker legs (rank 0, N = 1, after the timed region; oracle/ is loaded only here):
  parity        the C oracle builds the buckets of a read sample's features ITSELF from the same collection (mco_db_build) and
                classifies the sample; every candidate is compared with the GPU's.  configs[1]: the reference (oracle/_ref) on the
                database files this repo wrote.
  cpu_baseline  the same checker timed on the host cores, thread sweep, best value.

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

from metacache_amd import api, synth, synthdb  # noqa: E402
from metacache_amd.distributed import gather_candidates_async  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak (MI355X_MICROARCH.md)
READ_LEN = 150
PAD_LEN = 152                  # every read starts 4-byte aligned
KERNELS = ("plan", "sketch_probe", "sketch_lane", "chunk_sketch", "chunk_probe", "probe_cands", "mid_cands_64", "mid_cands_128",
           "mid_cands_256", "hash_cands_256", "hash_cands_512", "hash_cands_1024", "gw_filter_count", "gw_filter", "gw_filter2", "gw_compact", "gw_filter_stream_fine", "gw_filter_stream_mid", "gw_filter_stream", "gw_count", "gw_count_512", "gw_count_1024",
           "big_filter", "big_filter_2", "big_count", "big_count_2", "gw_sort", "gw_sorted_cands", "query_wave", "scan", "sort_candidates")
MULTI_KERNEL_TIMERS = ("plan", "scan", "gw_sort", "gw_compact", "gw_sorted_cands")   # timers over more than one kernel: never a line's `roofline.kernel`
KERNELS_MODE_K = ("mask_features", "gather_lists", "pack_numbers", "owner_entries", "decode_union", "cands_from_hits")   # shard / owner side of --mode K
# timer (mc_timing_get) -> the kernel's own name as the rocprofv3 summaries carry it (scripts/summarize_profile.py): prefixes
KERNEL_OF = {"sketch_lane": ("sketch_lane_kernel",), "probe_cands": ("probe_cands_kernel",), "sketch_probe": ("sketch_probe_lane_kernel",),
             "query_wave": ("query_kernel",), "sort_candidates": ("sort_candidates_kernel",),
             "gw_filter_count": ("gw_filter_count_kernel",), "gw_filter": ("gw_filter_kernel",),
             "gw_filter2": ("gw_filter2_kernel",), "gw_compact": ("gw_compact_kernel",), "gw_filter_stream_fine": ("gw_filter_stream_kernel<16",),
             "gw_filter_stream": ("gw_filter_stream_kernel<2u, 17",), "gw_filter_stream_mid": ("gw_filter_stream_kernel<2u, 16",), "gw_count": ("gw_count_kernel<9",), "gw_count_512": ("gw_count_kernel<10",),
             "gw_count_1024": ("gw_count_kernel<11",), "big_filter": ("big_filter_kernel",), "big_count": ("big_count_kernel<10",),
             "big_count_2": ("big_count_kernel<11",), "hash_cands_256": ("hash_cands_kernel<9",), "hash_cands_512": ("hash_cands_kernel<10",),
             "hash_cands_1024": ("hash_cands_kernel<11",), "mid_cands_64": ("mid_cands_kernel",), "mid_cands_128": ("mid_cands_kernel",),
             "mid_cands_256": ("mid_cands_kernel",), "gw_sort": ("gw_sort_chunk_kernel", "gw_sort_lists_kernel", "gw_merge_pass_kernel"), "gw_sorted_cands": ("gw_sorted_cands_kernel",),
             "gather_lists": ("gather_lists_kernel",), "owner_entries": ("owner_entries_kernel",), "decode_union": ("decode_union_kernel",)}
# ... and whole, for the kernels a line may name as its dominant one (profiles/r06*_kernel_stats.csv)
KERNEL_FULL = {"gw_filter_count": "gw_filter_count_kernel<4u, 14u, false, 7u>", "gw_filter2": "gw_filter2_kernel<4u, 14u>", "gw_filter_stream": "gw_filter_stream_kernel<2u, 17u, 15u, false>", "gw_filter_stream_mid": "gw_filter_stream_kernel<2u, 16u, 13u, false>",
               "gw_filter_stream_fine": "gw_filter_stream_kernel<16u, 19u, 17u, true>", "sketch_probe": "sketch_probe_lane_kernel<true>"}
# configs[2] at scale 1 (SURVEY §8d Config 3): 2000 genera x 4 species x 5 strains = 40 000 targets, 2.5 .. 5 Mbp each = 150 Gbp
CFG2 = dict(genera=2000, species_per_genus=4, strains_per_species=5, len_min=2_500_000, len_max=5_000_000, seed=3100)
# --shape refseq72k: the same 150 Gbp as a collection shaped like a real bacterial RefSeq -- 72 000 targets, most of 0.5 .. 3 Mbp, 3 % of
# the genera with genomes of 10 .. 16 Mbp (143 000 windows): target ids and window ids do NOT fit 32 bits together (17 + 18 bits), the
# global window numbers of the compact location store do
CFG2_72K = dict(genera=3600, species_per_genus=4, strains_per_species=5, len_min=500_000, len_max=3_000_000, seed=3100, big_fraction=0.03,
                big_len=(10_000_000, 16_000_000))


def make_genomes(n_genomes: int, length: int, seed: int):
    rng = np.random.default_rng(seed)
    return [synth.random_genome(rng, length) for _ in range(n_genomes)]


def synth_reads_gpu(gcat: torch.Tensor, goff: torch.Tensor, glen: int, n: int, seed: int) -> torch.Tensor:
    """configs[1]: n reads of READ_LEN on the GPU: uniform genome, uniform start, random strand, 1 % substitutions,
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


def make_long_reads(spec, gen, n: int, seed: int, dev, stride: int = 112):
    """BASELINE configs[4]'s reads (SURVEY §8d "Config 5"): n single reads, lengths log-normal around a median of 480 bp clipped to
    200 .. 19 000 (README.md:5), 7.5 % substitutions, 0.1 % N, both strands, uniform over the collection.  One packed character buffer
    (every read 4-byte aligned), qinfo rows {offset, length, offset, 0} and maxWindowsInRange = 2 + length / stride per read
    (candidate_structs.hpp:143-145).  Generated longest reads first, in groups with one row length."""
    rng = np.random.default_rng(seed)
    lens = np.clip(np.exp(rng.normal(np.log(480.0), 0.95, n)), 200, int(os.environ.get("MC_BENCH_LONG_CLIP", "19000"))).astype(np.int64)   # (the clip: experiments only)
    offs = np.zeros(n + 1, dtype=np.int64)
    offs[1:] = np.cumsum((lens + 3) // 4 * 4)
    if offs[-1] + 16 >= (1 << 32):
        sys.exit("--long-reads: batch too large (character offsets are 32 bits)")
    seq = torch.zeros(int(offs[-1]) + 16, dtype=torch.uint8, device=dev)
    order = np.argsort(-lens, kind="stable")
    lens_t = torch.from_numpy(lens).to(dev)
    offs_t = torch.from_numpy(offs[:-1].copy()).to(dev)
    done = 0
    while done < n:
        Lc = int(lens[order[done]])
        m = int(min(n - done, max(256, (96 << 20) // Lc)))
        sel = torch.from_numpy(order[done:done + m].copy()).to(dev)
        P = synthdb.read_params(spec, seed, read_len=Lc, sub_rate=0.075)
        rows = torch.zeros((m, P.row_bytes), dtype=torch.uint8, device=dev)
        gen.reads(spec, P, done, m, rows)
        pos = torch.arange(Lc, device=dev)[None, :]
        keep = pos < lens_t[sel][:, None]
        seq[(offs_t[sel][:, None] + pos)[keep]] = rows[:, :Lc][keep]
        done += m
    qinfo = torch.zeros((n, 4), dtype=torch.int32, device=dev)
    qinfo[:, 0] = offs_t.to(torch.int32); qinfo[:, 1] = lens_t.to(torch.int32); qinfo[:, 2] = qinfo[:, 0]
    maxwin = (2 + lens_t // stride).to(torch.int32)
    torch.cuda.synchronize()
    return {"seq": seq, "qinfo": qinfo, "maxwin": maxwin, "nchars": int(offs[-1]), "bases": int(lens.sum()), "lens": lens, "offs": offs}


def _pmc_rows(kernel_timer_name: str, tag: str):
    import csv
    fn = os.path.join(ROOT, "profiles", f"{tag}_pmc_summary.csv")
    if not os.path.exists(fn):
        return None, {}
    vals = {}
    for r in csv.DictReader(open(fn)):
        if any(r["kernel"].startswith(p) for p in KERNEL_OF.get(kernel_timer_name, ())):
            vals.setdefault(r["kernel"], {})[r["counter"]] = float(r["mean_per_dispatch"])
    return fn, vals


LINE_BYTES = 128.0            # what one read request of the L2's memory side costs against the HBM roofline (profiles/r05_fetch_calibration.md)


def pmc_layout_matches(tag: str, layout: dict):
    """profiles/<tag>_layout.json (scripts/summarize_profile.py: the table layout the PMC passes of that tag ran on, from the bench line of
    the same command) against the table of THIS run: a request count taken on another layout (lists on lines of their own or not, 4- or
    8-byte locations) says nothing about this one.  -> (ok, reason)"""
    fn = os.path.join(ROOT, "profiles", f"{tag}_layout.json")
    if not os.path.exists(fn):
        return False, f"profiles/{tag}_layout.json missing: the summary does not say which table layout it saw"
    rec = json.load(open(fn))
    for key, mine in (("table_location_bytes", layout["location_bytes"]), ("table_list_align", layout.get("list_align", 1)),
                      ("table_direct_index", bool(layout.get("direct_index", False)))):
        if rec.get(key) != mine:
            return False, f"profiles/{tag}_pmc_summary.csv was taken with {key} = {rec.get(key)}, this table has {mine}"
    return True, None


def measured_traffic(kernel_timer_name: str, tag: str):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 PMC summary of this workload
    (profiles/<tag>_pmc_summary.csv, separate --pmc passes): READ REQUESTS x 128 bytes + WRITE_SIZE.
    rocprofv3's FETCH_SIZE on gfx950 is TCC_EA0_RDREQ x 64 bytes whatever a request moves (MI355X_MICROARCH.md, HBM section); calibrated on this
    path's own access shapes (profiles/r05_fetch_calibration.md, tools/gather_width.hip): a request is one 128-byte LINE touched -- a list of 196
    bytes at a random 4-byte offset makes 2.6 requests for its 2.5 lines, not 4 for its 4 sectors -- and the memory system serves the same 47 x 10^9
    random requests per second for units of 32, 64 and 128 bytes and half as many units of 256: every request costs a line of the roofline.
    -> (bytes, file, detail) or (None, None, None) if no summary of this configuration exists."""
    fn, vals = _pmc_rows(kernel_timer_name, tag)
    for k, v in vals.items():
        if "TCC_EA0_RDREQ_sum" in v and "WRITE_SIZE" in v:
            rd = v["TCC_EA0_RDREQ_sum"] * LINE_BYTES
            return rd + v["WRITE_SIZE"] * 1024.0, os.path.basename(fn), {"read_requests": v["TCC_EA0_RDREQ_sum"], "bytes_per_request": LINE_BYTES,
                                                                          "WRITE_SIZE_bytes": v["WRITE_SIZE"] * 1024.0,
                                                                          "FETCH_SIZE_bytes_as_reported": v.get("FETCH_SIZE", 0.0) * 1024.0}
    return None, None, None


def gather_peak(gib: float):
    """the box's random-access peak (tools/gather_peak.hip): 64-byte requests per second for the three access shapes of the path, on
    a scratch buffer of `gib` GiB that is freed again before the database is built"""
    import ctypes
    from metacache_amd import build
    lib = ctypes.CDLL(build.build_gather_peak())
    lib.mcg_gather_peak.argtypes = [ctypes.c_uint64, ctypes.POINTER(ctypes.c_double)]
    r = (ctypes.c_double * 3)()
    if lib.mcg_gather_peak(int(gib * (1 << 30)), r) != 0:
        return None
    return {"buffer_GiB": gib, "lane_private_64B": r[0], "quad_64B": r[1], "wave_512B_list": r[2]}


def measured_requests(kernel_timer_name: str, tag: str):
    """TCC_EA0_RDREQ of the dominant kernel per launch from the committed PMC summary (64-byte requests, profiles/r01_fetch_calibration.md)"""
    _, vals = _pmc_rows(kernel_timer_name, tag)
    for v in vals.values():
        if "TCC_EA0_RDREQ_sum" in v:
            return v["TCC_EA0_RDREQ_sum"]
    return None


def kernel_bytes_per_read(timer: str, L: float, F: float, H: float, K: int, V: int):
    """the kernel's OWN share of SURVEY §8(d)'s bytes per read (ceil(L/4) + ceil(L/8) + 12 F + V H + 16 K): the read's characters belong to the
    sketching kernel, 12 F to the lookups, the V H bytes of the lists to the filter, the 16 K bytes of candidates to whoever writes them"""
    share = {"sketch_lane": (L + 3) // 4 + (L + 7) // 8, "sketch_probe": (L + 3) // 4 + (L + 7) // 8 + 12.0 * F, "probe_cands": 12.0 * F,
             "gw_filter_count": V * H + 16.0 * K, "gw_filter": V * H, "gw_filter2": V * H, "gw_filter_stream": V * H, "gw_filter_stream_fine": V * H, "gw_filter_stream_mid": V * H, "big_filter": V * H, "gw_count": 16.0 * K, "big_count": 16.0 * K,
             "gather_lists": V * H}
    return share.get(timer)


def algorithmic_bytes_per_read(F: float, H: float, K: int, V: int) -> float:
    """ceil(L/4) + ceil(L/8) + 12 F + V H + 16 K"""
    return (READ_LEN + 3) // 4 + (READ_LEN + 7) // 8 + 12.0 * F + V * H + 16.0 * K


def count_mismatches(gpu_cands: np.ndarray, cpu_cands: np.ndarray) -> int:
    mism = np.zeros(len(cpu_cands), dtype=bool)
    g = gpu_cands[:len(cpu_cands)]
    for f in ("tgt", "hits", "beg", "end"):
        mism |= ((g[f] != cpu_cands[f]) & ((g["hits"] > 0) | (cpu_cands["hits"] > 0))).any(axis=1)
    return int(mism.sum())


def thread_sweep(run, n_reads: int, budget_s: float, max_threads: int):
    """times run(n, threads) over a thread sweep inside the budget; -> (best Mreads/min, threads, {threads: Mreads/min})"""
    sweep, best = {}, (0.0, 1)
    probe_n = min(n_reads, 2000)
    t1, _ = run(probe_n, 1)
    rate1 = probe_n / max(t1, 1e-9)
    cands = [t for t in (1, 8, 16, 32, 64, 128, 256) if t <= max_threads]
    per = budget_s / max(len(cands), 1)
    for t in cands:
        n = int(min(n_reads, max(2000 * t, rate1 * min(t, 64) * per)))
        el, _ = run(n, t)
        v = n / max(el, 1e-9) * 60.0 / 1e6
        sweep[t] = round(v, 2)
        if v > best[0]:
            best = (v, t)
    return best[0], best[1], sweep


def cpu_leg_config1(dbname, reads_host, gpu_cands, K, budget_s):
    """configs[1]: the reference (oracle/_ref; the C oracle where it is absent) on the database files this repo wrote."""
    import cpuref
    n_total = reads_host.shape[0]
    kind = "reference" if cpuref.have_reference(2) else "port"
    ref = cpuref.reference(2) if kind == "reference" else cpuref.oracle()
    db = ref.open(dbname)

    def run(n, threads):
        seqs = np.ascontiguousarray(reads_host[:n, :READ_LEN]).reshape(-1)
        offs = np.arange(n + 1, dtype=np.uint64) * np.uint64(READ_LEN)
        return db.query_many(seqs, offs, max_cand=K, lowest=0, insert_max=0, threads=threads)

    import scale_util
    eff = scale_util.effective_cpus()                         # cgroup quota: more threads than that only take turns
    best, bt, sweep = thread_sweep(run, n_total, budget_s * 0.6, min(os.cpu_count() or 1, 4 * eff))
    n = int(min(n_total, max(100_000, best * 1e6 / 60.0 * budget_s * 0.4)))
    t, cands = run(n, bt)
    db.close()
    return ({"value": round(max(best, n / t * 60 / 1e6), 2), "unit": "Mreads/min", "cores": bt, "kind": kind, "thread_sweep": sweep,
             "host_cpus_granted": eff,
             "sample": f"{n} reads of the same workload (batch 0) on {bt} host threads (best of the sweep; the box grants {eff} CPUs), "
                       "database files written by this repo"},
            {"checked": n, "mismatches": count_mismatches(gpu_cands, cands), "against": kind})


def cpu_leg_reference_files(dbname, reads_host, gpu_cands, K, budget_s):
    """configs[2] with --reference-files: the reference itself loads the database files this run wrote and classifies the sample"""
    import cpuref
    import scale_util
    eff = scale_util.effective_cpus()
    t0 = time.time()
    db = cpuref.reference(4).open(dbname)
    load_s = time.time() - t0
    n = reads_host.shape[0]

    def run(m, threads):
        seqs = np.ascontiguousarray(reads_host[:m, :READ_LEN]).reshape(-1)
        offs = np.arange(m + 1, dtype=np.uint64) * np.uint64(READ_LEN)
        return db.query_many(seqs, offs, max_cand=K, lowest=0, insert_max=0, threads=threads)

    t, cands = run(n, min(os.cpu_count() or 1, 2 * eff))
    mism = count_mismatches(gpu_cands, cands)
    best, bt, sweep = thread_sweep(run, n, budget_s, min(os.cpu_count() or 1, 4 * eff))
    db.close()
    size = sum(os.path.getsize(dbname + e) for e in (".meta", ".cache0"))
    for e in (".meta", ".cache0"):
        os.remove(dbname + e)
    return ({"value": round(best, 2), "unit": "Mreads/min", "cores": bt, "kind": "reference", "thread_sweep": sweep, "host_cpus_granted": eff,
             "sample": f"{n} reads of the same workload (batch 0); the reference (oracle/_ref) on {bt} host threads (best of the sweep; the box grants {eff} "
                       f"CPUs) on the database files this run wrote ({size / 1e9:.0f} GB, loaded in {load_s:.0f} s)"},
            {"checked": n, "mismatches": mism, "against": "reference"})


def cpu_leg_config2(spec, reads_host, gpu_cands, K, n_parity, budget_s, mates_host=None):
    """configs[2]: a 100+ GB table is out of the checker's budget (the reference would need the whole database written to files and
    loaded single-threaded: minutes), but a read sample only ever looks at the buckets of ITS features: the C oracle builds exactly
    those from the same collection (mco_db_build: its own restatement of the database build) and classifies the sample."""
    import scale_util
    eff = scale_util.effective_cpus()                         # cgroup quota (the GPU boxes show 256 CPUs and grant 16)
    threads = min(os.cpu_count() or 1, 2 * eff)
    n = min(n_parity, reads_host.shape[0])
    t0 = time.time()
    sample = [reads_host[i, :READ_LEN].tobytes() for i in range(n)]
    if mates_host is not None:
        sample += [mates_host[i, :READ_LEN].tobytes() for i in range(n)]
    wanted = scale_util.sample_features(sample)
    odb = scale_util.oracle_database(spec, wanted, threads=threads)
    build_s = time.time() - t0
    if mates_host is not None:
        # pairs: the oracle's threaded bulk entry for pairs (mco_query_many_pairs: one query state per thread, a contiguous share of the pairs
        # each -- the reference's thread model); a pair counts as 2 reads
        def run_pairs(m, th):
            s1 = np.ascontiguousarray(reads_host[:m, :READ_LEN]).reshape(-1)
            s2 = np.ascontiguousarray(mates_host[:m, :READ_LEN]).reshape(-1)
            offs = np.arange(m + 1, dtype=np.uint64) * np.uint64(READ_LEN)
            return odb.query_many_pairs(s1, offs, s2, offs, max_cand=K, lowest=0, insert_max=0, threads=th)

        t, cands = run_pairs(n, threads)
        mism = count_mismatches(gpu_cands, cands)
        best, bt, sweep = thread_sweep(run_pairs, n, budget_s, min(os.cpu_count() or 1, 4 * eff))
        info = odb.info()
        odb.close()
        return ({"value": round(2 * best, 3), "unit": "Mreads/min", "cores": bt, "kind": "port", "thread_sweep_pairs_per_min": sweep, "host_cpus_granted": eff,
                 "sample": f"{n} pairs of the same workload (batch 0); C oracle (mco_query_many_pairs) on {bt} host threads (best of the sweep; the box grants {eff} "
                           f"CPUs); buckets of the sample's {len(wanted)} features ({info[7]} locations) built by the oracle itself in {build_s:.0f} s"},
                {"checked": n, "mismatches": mism, "against": "port (oracle builds its own buckets)"})

    def run(m, th):
        seqs = np.ascontiguousarray(reads_host[:m, :READ_LEN]).reshape(-1)
        offs = np.arange(m + 1, dtype=np.uint64) * np.uint64(READ_LEN)
        return odb.query_many(seqs, offs, max_cand=K, lowest=0, insert_max=0, threads=th)

    t, cands = run(n, threads)
    mism = count_mismatches(gpu_cands, cands)
    best, bt, sweep = thread_sweep(run, n, budget_s, min(os.cpu_count() or 1, 4 * eff))
    info = odb.info()
    odb.close()
    return ({"value": round(best, 2), "unit": "Mreads/min", "cores": bt, "kind": "port", "thread_sweep": sweep, "host_cpus_granted": eff,
             "sample": f"{n} reads of the same workload (batch 0); C oracle on {bt} host threads (best of the sweep; the box grants {eff} CPUs) against the buckets of the "
                       f"sample's {len(wanted)} features ({info[7]} locations), which it built itself from the same collection in {build_s:.0f} s "
                       f"on {threads} threads; the reference itself cannot load a table of this size inside the budget"},
            {"checked": n, "mismatches": mism, "against": "port (oracle builds its own buckets)"})


def cpu_leg_long_reads(spec, db, lb, K, n_parity, budget_s, rerun, out_cands):
    """--long-reads: a sample of the first batch against the C oracle, which builds the buckets of the sample's features itself (as
    cpu_leg_config2); the sample is bounded by its BASES (long reads collect tens of thousands of locations each)."""
    import scale_util
    eff = scale_util.effective_cpus()
    threads = min(os.cpu_count() or 1, 2 * eff)
    rerun()
    n = int(min(len(lb["lens"]), n_parity, max(200, np.searchsorted(np.cumsum(lb["lens"]), 16_000_000))))      # (round 6: 16 Mbases = 20 000 reads and more; 4 Mbases = 5 231 before)
    host = lb["seq"][: int(lb["offs"][n])].cpu().numpy()
    sample = [host[int(lb["offs"][i]): int(lb["offs"][i]) + int(lb["lens"][i])].tobytes() for i in range(n)]
    t0 = time.time()
    wanted = scale_util.sample_features(sample)
    odb = scale_util.oracle_database(spec, wanted, threads=threads)
    build_s = time.time() - t0
    gpu_c = out_cands[:n].cpu().numpy().view(np.uint32).reshape(n, K, 4)
    seqs = np.frombuffer(b"".join(sample), dtype=np.uint8)
    offs = np.zeros(n + 1, dtype=np.uint64)
    offs[1:] = np.cumsum(lb["lens"][:n])
    t, cands = odb.query_many(seqs, offs, max_cand=K, lowest=0, insert_max=0, threads=threads)
    gc = np.zeros((n, K), dtype=api.cand_dtype)
    gc["tgt"], gc["hits"], gc["beg"], gc["end"] = gpu_c[..., 0], gpu_c[..., 1], gpu_c[..., 2], gpu_c[..., 3]
    mism = count_mismatches(gc, cands)
    info = odb.info()
    odb.close()
    bases = int(lb["lens"][:n].sum())
    return ({"value": round(n / t * 60 / 1e6, 3), "unit": "Mreads/min", "cores": threads, "kind": "port", "host_cpus_granted": eff,
             "Gbases_per_s": round(bases / t / 1e9, 4),
             "sample": f"{n} reads ({bases} bases) of the same workload (batch 0); C oracle on {threads} host threads (the box grants {eff} CPUs) against the "
                       f"buckets of the sample's {len(wanted)} features ({info[7]} locations), which it built itself in {build_s:.0f} s"},
            {"checked": n, "mismatches": mism, "against": "port (oracle builds its own buckets)"})


def reference_calibration(scale: float, K: int, lf: float, device: int, budget_s: float):
    """what the port's figure means in terms of the REFERENCE.  The same collection at `scale` (a table the reference can
    load inside the budget) is built on the GPU, written as database files (mc_build_write_shards) to /dev/shm, loaded by oracle/_ref,
    and the same reads are classified by the reference on the whole table and by the port on buckets restricted to the reads' features
    (what the full-scale leg does) -- same threads.  Candidates of both are compared as well."""
    import cpuref
    import scale_util
    if not cpuref.have_reference(4):
        return None
    c2 = dict(CFG2); c2["genera"] = max(2, int(round(c2["genera"] * scale)))
    spec = synthdb.phylogeny(**c2)
    need = spec.total_bases // 112 * 16 * 9 * 2.5
    try:
        lim = open("/sys/fs/cgroup/memory.max").read().strip()
        lim = float("inf") if lim == "max" else float(lim)
    except OSError:
        lim = float("inf")
    shm = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    if need > 0.4 * lim or need > 100e9 or need > 0.8 * __import__("shutil").disk_usage(shm).free:
        return None
    name = os.path.join(shm, f"mcbench_cal_{os.getpid()}")
    eff = scale_util.effective_cpus()
    threads = min(os.cpu_count() or 1, 2 * eff)
    try:
        shards = max(1, int(np.ceil(spec.total_bases // 112 * 16 / 1.4e9)))
        t0 = time.time()
        db, _ = synthdb.build_database(spec, device=device, shards=shards, max_candidates=K, max_load_factor=lf, write_to=name)
        n = 100_000
        P = synthdb.read_params(spec, 3100)
        rows = torch.zeros((n, P.row_bytes), dtype=torch.uint8, device=torch.device("cuda", device))
        synthdb.GpuSynth(device).reads(spec, P, 0, n, rows)
        gpu, counts, _ = db.query([bytes(r[:READ_LEN]) for r in rows.cpu().numpy()][:20_000])
        locs = int(db.info()[7])
        db.close()
        build_s = time.time() - t0
        reads_host = rows.cpu().numpy()
        seqs = np.ascontiguousarray(reads_host[:, :READ_LEN]).reshape(-1)
        offs = np.arange(n + 1, dtype=np.uint64) * np.uint64(READ_LEN)
        t0 = time.time()
        ref = cpuref.reference(4).open(name)
        load_s = time.time() - t0
        ref.query_many(seqs[: 2000 * READ_LEN], offs[:2001], max_cand=K, lowest=0, insert_max=0, threads=threads)
        t_ref, c_ref = ref.query_many(seqs, offs, max_cand=K, lowest=0, insert_max=0, threads=threads)
        ref.close()
        sample = [reads_host[i, :READ_LEN].tobytes() for i in range(n)]
        odb = scale_util.oracle_database(spec, scale_util.sample_features(sample), threads=threads)
        odb.query_many(seqs[: 2000 * READ_LEN], offs[:2001], max_cand=K, lowest=0, insert_max=0, threads=threads)
        t_port, c_port = odb.query_many(seqs, offs, max_cand=K, lowest=0, insert_max=0, threads=threads)
        odb.close()
        return {"scale": scale, "db_bases": int(spec.total_bases), "db_locations": locs, "reads": n, "threads": threads,
                "locations_per_read": round(float(counts.mean()), 1),
                "port_Mreads_min": round(n / t_port * 60 / 1e6, 2), "reference_Mreads_min": round(n / t_ref * 60 / 1e6, 2),
                "port_over_reference": round(t_ref / t_port, 3),
                "reference_vs_port_mismatches": count_mismatches(c_ref, c_port), "reference_vs_gpu_mismatches": count_mismatches(gpu, c_ref[:len(gpu)]),
                "reference_load_s": round(load_s, 1), "build_and_write_s": round(build_s, 1)}
    finally:
        for e in (".meta", ".cache0"):
            if os.path.exists(name + e):
                os.remove(name + e)


def host_fed_leg(db, batches, long_batches, qinfo, nloc, nchars, max_win, K, steps, warmup, nb, per_read, kernel_only_s):
    """SURVEY 8(d) row 1, the host-fed form: the timed region's K batches once more, this time starting in PINNED HOST memory -- the upload
    of batch i + 1 on an upload stream of its own under the kernels of batch i, the candidates copied out of the pipe's buffer on the device
    and back to pinned host memory on a download stream (mc_copy_results_on kind 0, then kind 1) inside the clock.  What a host application that parses reads itself can reach at most:
    never `value`."""
    dev = qinfo.device
    srcs = [(long_batches[(warmup + i) % nb] if long_batches is not None else batches[(warmup + i) % nb]) for i in range(steps)]
    t0 = time.perf_counter()
    host_in = []
    for sidx in range(steps):
        t = srcs[sidx]["seq"] if long_batches is not None else srcs[sidx]
        h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        h.copy_(t)
        host_in.append(h)
    nbytes_in = [int(h.numel()) for h in host_in]
    # THREE input buffers on the device and an upload stream of its own: the upload of batch i + 1 is enqueued as soon as batch i's kernels are
    # -- it must not sit behind the tail of batch i - 1 on that batch's stream (a few small kernels that wait for CUs while batch i's
    # kernels fill the device: behind them the upload started a batch late, 30 ms per step instead of 18)
    dev_in = [torch.zeros(max(nbytes_in) + 16, dtype=torch.uint8, device=dev) for _ in range(3)]
    host_out = [torch.zeros((nloc, K, 4), dtype=torch.int32).pin_memory() for _ in range(2)]
    # the candidates leave the pipe's own buffer by a copy on the device (0.1 ms) and go to the host on a DOWNLOAD stream: 0.16 GB at the
    # link's rate are 3 ms, which on the pipe's stream stood in front of the pipe's next batch
    dev_out = [torch.zeros((nloc, K, 4), dtype=torch.int32, device=dev) for _ in range(2)]
    down = torch.cuda.Stream(device=dev)
    out_ready = [torch.cuda.Event() for _ in range(2)]
    out_free = [torch.cuda.Event() for _ in range(2)]
    out_used = [False, False]
    # the upload in NUP shares on as many streams (copy engines): MC_BENCH_UP_STREAMS, default 2
    NUP = max(1, int(os.environ.get("MC_BENCH_UP_STREAMS", "1")))   # (measured: 1 -> 21.0, 2 -> 22.1, 3 -> 26.1 ms per step; one share runs at the link's 57 GB/s under the kernels)
    ups = [torch.cuda.Stream(device=dev) for _ in range(NUP)]
    up = ups[0]
    pipes = [torch.cuda.Stream(device=dev) for _ in range(2)]
    up_done = [[torch.cuda.Event() for _ in range(NUP)] for _ in range(3)]
    in_free = [torch.cuda.Event() for _ in range(3)]
    up_t0 = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]     # how long an upload takes UNDER the kernels (share 0's stream)
    up_t1 = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
    torch.cuda.synchronize()
    stage_s = time.perf_counter() - t0
    pend = {}
    trace = {"h2d_call": 0.0, "query_call": 0.0, "finish_call": 0.0, "d2h_call": 0.0}

    def clocked(key, f, *a, **kw):
        t = time.perf_counter()
        r = f(*a, **kw)
        trace[key] += time.perf_counter() - t
        return r

    def upload(i):
        k = i % 3
        share = (nbytes_in[i] + NUP - 1) // NUP // 256 * 256 + 256
        for u, st in enumerate(ups):
            lo, hi = min(nbytes_in[i], u * share), min(nbytes_in[i], (u + 1) * share)
            with torch.cuda.stream(st):
                if i >= 3:
                    st.wait_event(in_free[k])                # batch i - 3 has read this buffer to its end (main kernels and tail)
                if u == 0:
                    up_t0[i].record(st)
                if hi > lo:
                    dev_in[k][lo:hi].copy_(host_in[i][lo:hi], non_blocking=True)
                if u == 0:
                    up_t1[i].record(st)
                up_done[k][u].record(st)

    def finish_pipe(j):
        if j not in pend:
            return
        ptr, i = pend.pop(j)
        clocked("finish_call", db.query_finish, second_pipe=bool(j))
        in_free[i % 3].record(pipes[j])
        if out_used[j]:
            pipes[j].wait_event(out_free[j])                 # (the download of this pipe's batch before last has left dev_out[j])
        clocked("d2h_call", db.copy_results, dev_out[j].data_ptr(), ptr, nloc * K * 16, second_pipe=bool(j), stream=pipes[j].cuda_stream)
        out_ready[j].record(pipes[j])
        down.wait_event(out_ready[j])
        clocked("d2h_call", db.copy_results, host_out[j].data_ptr(), dev_out[j].data_ptr(), nloc * K * 16, to_host=True, second_pipe=bool(j), stream=down.cuda_stream)
        out_free[j].record(down)
        out_used[j] = True

    def run():
        db.synchronize(); torch.cuda.synchronize()
        for k in trace:
            trace[k] = 0.0
        t0 = time.perf_counter()
        clocked("h2d_call", upload, 0)
        for i in range(steps):
            j = i & 1
            if i + 1 < steps:
                clocked("h2d_call", upload, i + 1)
            for e in up_done[i % 3]:
                pipes[j].wait_event(e)
            src = dev_in[i % 3]
            if long_batches is not None:
                lb = srcs[i]
                res = clocked("query_call", db.query_device, src.data_ptr(), lb["qinfo"].data_ptr(), nloc, lb["nchars"], max_win_ptr=lb["maxwin"].data_ptr(), second_pipe=bool(j),
                              defer_tail=True, stream=pipes[j].cuda_stream)
            else:
                res = clocked("query_call", db.query_device, src.data_ptr(), qinfo.data_ptr(), nloc, nchars, max_win_uniform=max_win, second_pipe=bool(j), defer_tail=True,
                              stream=pipes[j].cuda_stream)
            pend[j] = (res.cands, i)
            finish_pipe(j ^ 1)
        finish_pipe(0); finish_pipe(1)
        torch.cuda.synchronize()
        db.synchronize()
        return time.perf_counter() - t0

    run()                                                    # (first touch of the pinned buffers by the device)
    el = min(run(), run())
    up_ms = sorted(up_t0[i].elapsed_time(up_t1[i]) for i in range(min(1, steps - 1), steps))   # (the last run's; share 0 of NUP; without the first, which has nothing beside it)
    up_gb = sum(nbytes_in) / 1e9
    down = steps * nloc * K * 16 / 1e9
    # the link alone: the same uploads back to back, nothing else on the device
    db.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        with torch.cuda.stream(up):
            dev_in[i % 3][: nbytes_in[i]].copy_(host_in[i], non_blocking=True)
    torch.cuda.synchronize()
    link_s = time.perf_counter() - t0
    ms = el / steps * 1e3
    ratio = kernel_only_s / (el / steps)
    out = {"ms_per_step": round(ms, 3), "Mreads_min": round(steps * nloc * per_read / el * 60 / 1e6, 1), "h2d_GBps": round(up_gb / link_s, 1),
           "h2d_GB_per_step": round(up_gb / steps, 3), "d2h_GB_per_step": round(down / steps, 4), "over_kernel_only": round(ratio, 3),
           "upload_streams": NUP, "h2d_ms_of_one_share_under_the_kernels_median": round(up_ms[len(up_ms) // 2], 2),
           "h2d_GBps_under_the_kernels": round(up_gb / steps / NUP / (up_ms[len(up_ms) // 2] / 1e3), 1),
           "staging_s": round(stage_s, 1), "host_ms_per_step_inside_the_calls": {k: round(v / steps * 1e3, 3) for k, v in trace.items()},
           "what": "the timed region's batches from pinned host memory: H2D of batch i+1 on an upload stream under batch i's kernels, candidates copied out of the pipe's buffer on the device and D2H to pinned memory on a download stream, all inside the clock; best of 2"}
    if ratio < 0.8:
        h2d_ms = link_s / steps * 1e3
        out["bound"] = (f"PCIe: the upload alone takes {h2d_ms:.1f} ms per step at {up_gb / link_s:.0f} GB/s" if h2d_ms > 0.8 * ms else
                        f"neither the link ({h2d_ms:.1f} ms per step) nor the kernels ({kernel_only_s * 1e3:.1f}): the copies and the kernels do not overlap fully")
    del host_in, dev_in, host_out, dev_out
    return out


def e2e_leg(scale: float, budget_s: float) -> dict:
    """end-to-end `mcq query` wall time on database FILES -- tools/e2e_scale.py as its own process once this one has
    given the device back: the collection written to /dev/shm by the streaming writer, 10^7 reads in a FASTA file, `mcq query -no-map`
    and `-tophits -queryids`.  File reading, parsing, PCIe, classification and printing included: never `value`."""
    import shutil
    import signal
    import subprocess
    shm = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    try:
        lim = open("/sys/fs/cgroup/memory.max").read().strip()
        lim = float("inf") if lim == "max" else float(lim)
        cur = float(open("/sys/fs/cgroup/memory.current").read().strip())
    except (OSError, ValueError):
        lim, cur = float("inf"), 0.0
    spec_bytes = lambda sc: 150.5e9 * sc / 112 * 16 * 8.2 + 2.0e9          # noqa: E731  (8 bytes per location + keys and sizes; + the FASTA file)
    note = None
    for sc in ([scale] + ([0.2] if scale > 0.2 else [])):
        need = spec_bytes(sc)
        if need * 1.25 + 30e9 < min(lim - cur, shutil.disk_usage(shm).free):
            scale = sc
            break
        note = f"scale {sc}: {need / 1e9:.0f} GB of files do not fit this box's memory allowance ({(lim - cur) / 1e9:.0f} GB left, {shutil.disk_usage(shm).free / 1e9:.0f} GB of {shm})"
    else:
        return {"skipped": note}
    out = os.path.join(tempfile.gettempdir(), f"mc_e2e_{os.getpid()}.json")
    cmd = [sys.executable, os.path.join(ROOT, "tools", "e2e_scale.py"), "--scale", str(scale), "--no-ref", "--runs", "mcq_nomap,mcq_tophits_ids", "--out", out]
    t0 = time.time()
    p = subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, start_new_session=True, cwd=ROOT)
    try:
        _, err = p.communicate(timeout=budget_s)
    except subprocess.TimeoutExpired:
        os.killpg(p.pid, signal.SIGKILL)                       # exactly the process group started above
        p.communicate()
        for f in os.listdir(shm):
            if f.startswith(f"mc_e2e_{p.pid}"):
                os.remove(os.path.join(shm, f))
        return {"skipped": f"time limit of {budget_s:.0f} s reached", "scale": scale}
    if p.returncode != 0 or not os.path.exists(out):
        return {"skipped": f"exit code {p.returncode}: " + (err or b"").decode(errors="replace")[-300:], "scale": scale}
    r = json.load(open(out))
    os.remove(out)
    res = {"scale": scale, "reads": r["reads"], "database_file_GB": round(r["collection"]["database_file_bytes"] / 1e9, 1), "seconds": round(time.time() - t0, 1)}
    for k in ("mcq_nomap", "mcq_tophits_ids"):
        if k in r:
            res[k] = {"query_ms": r[k]["query_ms"], "Mreads_min": r[k]["Mreads_per_min_query_phase"], "database_load_and_startup_s": r[k]["database_load_and_startup_s"],
                      "wall_s": r[k]["wall_s"]}
    if note:
        res["note"] = note
    return res


def run_multi_gpu_selfcheck(world: int, budget_s: float) -> dict:
    """After the timed region: tools/multi_gpu_selfcheck.py as its OWN job over the same N GPUs (its own rendezvous, its own processes, a
    hard time limit: nothing in it can hang or fail this run) -- modes P and K across N real ranks and the C++ mc_keyset / mc_partset
    drivers over all N devices, every candidate against the oracle.  Returns what it printed, or what went wrong."""
    import signal
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = os.path.join(tempfile.gettempdir(), f"mc_selfcheck_{os.getpid()}.json")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "ROLE_NAME",
                                                              "GROUP_WORLD_SIZE", "ROLE_WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR", "TORCHELASTIC_RUN_ID",
                                                              "TORCHELASTIC_RESTART_COUNT", "TORCHELASTIC_MAX_RESTARTS", "TORCHELASTIC_USE_AGENT_STORE",
                                                              "TORCHELASTIC_ERROR_FILE", "TORCH_NCCL_ASYNC_ERROR_HANDLING", "OMP_NUM_THREADS")}
    env["MASTER_ADDR"] = "127.0.0.1"; env["MASTER_PORT"] = str(port)
    tool = os.path.join(ROOT, "tools", "multi_gpu_selfcheck.py")
    if world > 1:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1", "--master-port", str(port),
               tool, "--out", out]
    else:
        cmd = [sys.executable, tool, "--out", out]
    t0 = time.time()
    try:
        p = subprocess.Popen(cmd, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, start_new_session=True, cwd=ROOT)
        try:
            _, err = p.communicate(timeout=budget_s)
        except subprocess.TimeoutExpired:
            os.killpg(p.pid, signal.SIGKILL)                   # exactly the process group started above
            p.communicate()
            return {"ranks_seen": world, "ok": False, "error": f"time limit of {budget_s:.0f} s reached", "seconds": round(time.time() - t0, 1)}
        if os.path.exists(out):
            res = json.loads(open(out).read())
            os.remove(out)
            res["exit_code"] = p.returncode
            return res
        return {"ranks_seen": world, "ok": False, "error": f"exit code {p.returncode}: " + (err or b"").decode(errors="replace")[-400:], "seconds": round(time.time() - t0, 1)}
    except Exception as e:                                      # noqa: BLE001  (never fatal for the headline number)
        return {"ranks_seen": world, "ok": False, "error": repr(e)}


def self_launch_command(argv, gpus, port):
    """the command `python bench.py --gpus N <args>` re-executes itself as when it is not already a rank of a process group"""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(argv, gpus):
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = self_launch_command(argv, gpus, port)
    print("[bench] " + " ".join(cmd), file=sys.stderr, flush=True)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=2, choices=(1, 2), help="BASELINE.json configs[i]")
    ap.add_argument("--batch", type=int, default=0, help="reads (pairs) per step per GPU (default: configs[2] 5 M reads, 2.5 M pairs, 250 000 long reads; configs[1] 10 M)")
    ap.add_argument("--scale", type=float, default=1.0, help="configs[2]: fraction of the 2000 genera (quick runs)")
    ap.add_argument("--build-shards", type=int, default=0, help="configs[2]: key-shard passes of the build (0 = by size)")
    ap.add_argument("--shape", default="default", choices=("default", "refseq72k"), help="configs[2]: the collection's shape (refseq72k: 72 000 targets, "
                    "genomes up to 16 Mbp, the same 150 Gbp)")
    ap.add_argument("--genomes", type=int, default=16)
    ap.add_argument("--genome-len", type=int, default=5_000_000)
    ap.add_argument("--maxcand", type=int, default=2)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU baseline leg (0 = skip both checker legs)")
    ap.add_argument("--parity-reads", type=int, default=200_000, help="configs[2]: reads checked against the oracle")
    ap.add_argument("--load-factor", type=float, default=0.0, help="0 = 0.3 for configs[1], 0.5 for configs[2]")
    ap.add_argument("--gather-gib", type=float, default=-1.0, help="scratch buffer of the random-access microbenchmark (0 = skip; default: 64 for "
                    "configs[2], 0.6 = the table's size for configs[1])")
    ap.add_argument("--mode", default="R", choices=("R", "P", "K"), help="configs[2]: R replicated table (default), P one part per rank, K key shards")
    ap.add_argument("--wire", type=int, default=4, choices=(4, 8), help="mode K: bytes per location in the exchange (4 = global window numbers, 8 = (target, window))")
    ap.add_argument("--pairs", action="store_true", help="configs[2]: 2 x 150 bp read pairs (configs[3]'s reads) instead of single reads")
    ap.add_argument("--reference-files", default="", help="configs[2], N = 1: also write the database as files under this name (e.g. /dev/shm/mcdb: "
                    "190 GB at full scale) and let the REFERENCE (oracle/_ref) load them and be the checker and the CPU baseline instead of the oracle")
    ap.add_argument("--force-dist", action="store_true", help="run the N>1 gather path with a single rank too (testing)")
    ap.add_argument("--no-pipeline", action="store_true", help="mode R: one batch at a time on one pipe (default: two batches in flight on the context's two "
                    "pipes, mc_query_device(MC_DEFER_TAIL) + mc_query_finish)")
    ap.add_argument("--selfcheck-seconds", type=float, default=150.0, help="time limit of the multi-GPU self-check that follows the timed region "
                    "(tools/multi_gpu_selfcheck.py as its own job over the same N ranks; 0 = skip)")
    ap.add_argument("--tune", default="", help="run-time tuning switches (mc_set_tuning) for experiments: name=value[,name=value ...], e.g. gw_fuse=0")
    ap.add_argument("--repeats", type=int, default=3, help="timed repeats of the K steps: the first is the line's value, all of them its value_range")
    ap.add_argument("--long-reads", action="store_true", help="configs[2] table, BASELINE configs[4]'s reads: single reads of 200 .. 19 000 bp (log-normal, "
                    "median 480), 7.5 %% substitutions, seed 5100; --batch = reads per step (default 250 000)")
    ap.add_argument("--calibrate-scale", type=float, default=0.1, help="configs[2], N = 1: after the run, the same collection at this scale is built, written "
                    "as database files and classified by the REFERENCE (oracle/_ref) and by the port on the same reads: cpu_baseline.reference_calibration "
                    "(0 = skip)")
    ap.add_argument("--host-fed", type=int, default=1, help="N = 1, mode R: after the timed region the same K batches once more from PINNED HOST memory through the "
                    "two pipes, H2D of the reads and D2H of the candidates inside the clock (`host_fed` in the line; SURVEY 8d: reads pre-staged in pinned host memory); 0 = skip")
    ap.add_argument("--e2e-scale", type=float, default=1.0, help="N = 1, default workload: after everything else tools/e2e_scale.py as its own process -- the collection at this "
                    "scale as database FILES in /dev/shm, 10^7 reads as a FASTA file, `mcq query -no-map` and `-tophits -queryids` (`e2e` in the line); falls back to 0.2 "
                    "where the box's memory allowance does not hold the full-scale file set; 0 = skip")
    ap.add_argument("--e2e-seconds", type=float, default=330.0, help="time limit of that process (it takes 140-200 s at full scale: 100 s to build and write the 174 GB file set, "
                    "20 + 10 s of loading for the two mcq runs)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (args.gpus > 1 or args.force_dist) and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU over RCCL) and hand on their exit code;
        # rank 0's JSON line is the last line of stdout either way
        sys.exit(self_launch(sys.argv[1:], args.gpus))
    if args.gpus != world:
        sys.exit(f"bench.py --gpus {args.gpus} inside a process group of {world} ranks: start one rank per GPU (--nproc-per-node = --gpus)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1 or args.force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    if args.gather_gib < 0:
        args.gather_gib = 64.0 if args.config == 2 else 0.6
    peak = gather_peak(args.gather_gib) if rank == 0 and args.gather_gib > 0 else None
    K = args.maxcand
    cfg = args.config
    B = args.batch or ((250_000 if args.long_reads else 2_500_000 if args.pairs else 5_000_000) if cfg == 2 else 10_000_000)   # (5 M PAIRS ask for 2^32 pool entries per pipe: beyond the device)
    if args.long_reads and (cfg != 2 or args.mode != "R" or args.pairs):
        sys.exit("--long-reads: configs[2] table, mode R, single reads")
    lf = args.load_factor or 0.3                             # configs[2]: 0.5 -> 0.3 is 5.15 -> 4.55 ms of probing per 5 M reads for 10 GB more buckets
    # distinct batches resident in HBM: configs[2] in mode R one per step, warm-up included (25 x 0.76 GB: no timed batch has been
    # through the path before); configs[1] and the sharded modes (N x B reads per rank and step) cycle through a few
    if cfg == 2 and args.mode == "R" and not args.long_reads:
        nb = max(1, min(args.steps + args.warmup, 48))
    else:
        nb = max(1, min(max(args.steps, args.warmup), 8 if cfg == 1 else 4))
    spec = None
    dbdir = None
    build_info = {}
    t0 = time.time()
    if cfg == 1:
        # ---- configs[1]: 16 x 5 Mbp, uint16 targets, replicated on every rank, built on the GPU by our builder ------------------
        genomes = make_genomes(args.genomes, args.genome_len, seed=16)
        bld = api.Builder(device=local, target_id_bytes=2, max_candidates=K, max_load_factor=lf)
        for i, gnm in enumerate(genomes):
            bld.add_target(gnm, f"SYN_{i:06d}.1", parent_taxid=1000 + i, filename=f"syn{i}.fa")
        db = bld.finish(load=True)
        if rank == 0 and args.cpu_seconds > 0 and world == 1:
            dbdir = tempfile.mkdtemp(prefix="mcbench")
            taxa = [(1, 1, 20, "root")] + [(1000 + i, 1, 4, f"synthetic species {i}") for i in range(args.genomes)]
            bld.write(os.path.join(dbdir, "syn16"), taxa)
        bld.free()
        gcat = torch.from_numpy(np.concatenate(genomes)).to(dev)
        goff = torch.arange(args.genomes, device=dev, dtype=torch.int64) * args.genome_len
        batches = [synth_reads_gpu(gcat, goff, args.genome_len, B, seed=1016 + 7919 * rank + s).reshape(-1) for s in range(nb)]
        del gcat
        V = 6                                                  # uint16 target ids: 6-byte locations in the file format
        workload = (f"configs[1]: {args.genomes}x{args.genome_len} bp synthetic DB (uint16 target ids, 1 partition), "
                    f"{world * args.steps * B} synthetic 150 bp reads")
        pmc_tag = "r02c1"
    else:
        # ---- configs[2]: RefSeq-scale phylogeny, uint32 targets, built on this GPU in key shards ---------------------------------
        c2 = dict(CFG2 if args.shape == "default" else CFG2_72K)
        c2["genera"] = max(2, int(round(c2["genera"] * args.scale)))
        spec = synthdb.phylogeny(**c2)
        est_pairs = spec.total_bases // 112 * 16
        shards = args.build_shards or max(1, int(np.ceil(est_pairs / 1.4e9)))
        say = (lambda m: print("[bench] " + m, file=sys.stderr, flush=True)) if rank == 0 else None
        mode = args.mode
        part_sel = None
        if mode == "P":                                        # part r = targets r, r + N, r + 2N, ... (round-robin like the reference's -parts)
            part_sel = np.arange(rank, len(spec.targets), world)
            pspec = spec.subset(part_sel)
            pshards = max(1, int(np.ceil(pspec.total_bases // 112 * 16 / 1.4e9)))
            db, build_info = synthdb.build_database(pspec, device=local, shards=pshards, max_candidates=K, max_load_factor=lf, report=say)
        elif mode == "K":
            kshards = max(1, int(np.ceil(shards / world)))
            db, build_info = synthdb.build_database(spec, device=local, shards=kshards, key_shard=(rank, world), max_candidates=K, max_load_factor=lf,
                                                    report=say)
        else:
            write_to = (args.reference_files if (rank == 0 and world == 1) else "") or None
            if write_to:
                # the files (and the reference's copy of them in RAM) must fit the memory this process group is granted: a full-scale set
                # (190 GB, twice) does not fit the GPU boxes' allowance -- that run took the box down; mid scale (19 GB) is fine
                need = spec.total_bases // 112 * 16 * 9 * 2.5
                try:
                    lim = open("/sys/fs/cgroup/memory.max").read().strip()
                    lim = float("inf") if lim == "max" else float(lim)
                except OSError:
                    lim = float("inf")
                if need > 0.5 * lim or need > 120e9:
                    sys.exit(f"--reference-files: about {need / 1e9:.0f} GB of files + reference tables do not fit this box's memory allowance; use --scale <= 0.2")
            db, build_info = synthdb.build_database(spec, device=local, shards=shards, max_candidates=K, max_load_factor=lf, report=say, write_to=write_to)
        gen = synthdb.GpuSynth(local)
        if args.long_reads:
            long_batches = [make_long_reads(spec, gen, B, 5100 + 7919 * rank + sidx, dev) for sidx in range(nb)]
        P = synthdb.read_params(spec, 4100 if args.pairs else 3100, paired=args.pairs)
        assert P.row_bytes == PAD_LEN
        # modes P and K: every rank works on ALL reads of a step (N x B); mode R: on its own B
        nloc = B * world if mode in ("P", "K") else B
        batches, mates = [], []
        for sidx in range(0 if args.long_reads else nb):
            t = torch.zeros(nloc * PAD_LEN, dtype=torch.uint8, device=dev)
            t2 = torch.zeros(nloc * PAD_LEN + 16, dtype=torch.uint8, device=dev) if args.pairs else None
            first = sidx * nloc if mode in ("P", "K") else (rank * 64 + sidx) * B
            gen.reads(spec, P, first, nloc, t, t2)
            batches.append(t); mates.append(t2)
        V = 8
        max_len = int(spec.targets["length"].max())
        shape = "2 x 150 bp read pairs" if args.pairs else "150 bp reads"
        how = {"R": "1 partition", "P": f"{world} partitions (targets round-robin), one per GPU", "K": f"1 partition key-sharded over {world} GPUs"}[mode]
        workload = (f"configs[{4 if args.long_reads else 3 if (args.pairs or mode != 'R') else 2}]: RefSeq-scale synthetic DB, {len(spec.targets)} targets / {spec.total_bases / 1e9:.1f} Gbp "
                    f"(genus>species>strain phylogeny{'' if args.shape == 'default' else f', genomes up to {max_len / 1e6:.1f} Mbp'}, uint32 target ids, {how}{'' if args.scale == 1.0 else f', scale {args.scale}'}), "
                    + (f"{world * args.steps * B} synthetic long reads (200-19000 bp, log-normal, median 480; 7.5 % substitutions)" if args.long_reads else
                       f"{world * args.steps * B * (2 if args.pairs else 1)} synthetic {shape}")
                    + ("" if nb >= args.steps + args.warmup else f" ({world * nb * B * (2 if args.pairs else 1)} distinct, cycled)"))
        # the committed PMC passes (profiles/r04_pmc_summary.csv) ran the default command: full scale, 5 M reads per step, mode R
        # the committed PMC passes (scripts/profile.sh; profiles/r06*_pmc_summary.csv + _layout.json) ran these commands: full scale, mode R, default batch sizes
        pmc_tag = ("r06" if (not args.pairs and not args.long_reads and B == 5_000_000) else "r06_long" if (args.long_reads and B == 250_000) else "r06_pairs" if (args.pairs and B == 2_500_000) else "r06-none") if (mode == "R" and args.scale == 1.0) else "r06-none"
    build_s = time.time() - t0
    db_info = db.info()
    for kv in filter(None, args.tune.split(",")):
        db.set_tuning(kv.split("=")[0], int(kv.split("=")[1]))

    mode = args.mode if cfg == 2 else "R"
    pairs = bool(args.pairs) and cfg == 2
    nloc = B * world if mode in ("P", "K") else B               # reads (pairs) this rank looks up per step
    qinfo = torch.zeros((nloc, 4), dtype=torch.int32, device=dev)
    qinfo[:, 0] = torch.arange(nloc, device=dev, dtype=torch.int32) * PAD_LEN
    qinfo[:, 1] = READ_LEN
    qinfo[:, 2] = qinfo[:, 0]
    slack = torch.zeros(16, dtype=torch.uint8, device=dev)
    if pairs:
        # one character buffer per batch: mate-1 rows, then mate-2 rows
        batches = [torch.cat([b1, m2]) for b1, m2 in zip(batches, mates)]
        qinfo[:, 2] = qinfo[:, 0] + nloc * PAD_LEN
        qinfo[:, 3] = READ_LEN
    else:
        batches = [torch.cat([b, slack]) for b in batches]
    nchars = nloc * PAD_LEN * (2 if pairs else 1)
    max_win = db.max_windows_in_range(READ_LEN, READ_LEN if pairs else 0)      # 3 for a 150 bp read, 4 for a 2 x 150 pair
    part_sel_t = torch.from_numpy(part_sel).to(dev).to(torch.int32) if (cfg == 2 and mode == "P") else None
    # N > 1: the per-rank candidate lists go to rank 0 over RCCL (north_star: "per-rank partial hit lists gathered over RCCL/xGMI
    # before host-side taxonomy assignment").  Two buffers per rank: the gather of batch i runs while batch i+1 is computed.
    dist_path = world > 1 or args.force_dist
    pipelined = mode == "R" and not args.no_pipeline
    nbuf = 2 if (dist_path or pipelined) else 1
    nout = B if mode in ("R", "K") else nloc                    # candidate rows this rank ends up with per step
    out_bufs = [torch.zeros((nout, K, 4), dtype=torch.int32, device=dev) for _ in range(nbuf)]
    out_cands = out_bufs[0]
    recv = [[torch.zeros((B, K, 4), dtype=torch.int32, device=dev) for _ in range(world)] for _ in range(nbuf)] if dist_path and rank == 0 and mode != "P" else None
    works = [None] * nbuf
    torch.cuda.synchronize()
    from metacache_amd.distributed import classify_key_sharded_device, classify_partitioned
    numbers_wire = mode == "K" and args.wire == 4 and db.table_layout()["location_bytes"] == 4     # the lists leave the shard as the 4-byte numbers they are stored as

    def finish(j: int):
        if works[j] is not None:
            works[j].wait()
            torch.cuda.current_stream().synchronize()
            works[j] = None

    pend = {}                                                # pipe -> device pointer of the candidates its batch in flight will leave

    def finish_pipe(j: int):
        """mode R, two batches in flight: the tail of the batch on pipe j (mc_query_finish: the host looks at its counters only now, with
        the other pipe's batch queued behind it), its candidates into out_bufs[j], then (N > 1) their gather to rank 0"""
        if j not in pend:
            return
        ptr = pend.pop(j)
        db.query_finish(second_pipe=bool(j))
        db.copy_results(out_bufs[j].data_ptr(), ptr, nloc * K * 16, second_pipe=bool(j))
        if dist_path:
            db.query_wait(second_pipe=bool(j))               # RCCL reads the buffer on torch's stream
            works[j] = gather_candidates_async(out_bufs[j], recv[j] if recv is not None else None, dst=0)

    first_batch = [0]                                        # warm-up takes batches 0 .. W-1, the timed region W .. W+K-1: no timed batch has been through the path before

    def step(i: int):
        i += first_batch[0]
        b = batches[i % nb] if batches else None
        j = i % nbuf
        finish(j)                                            # the gather that used this buffer two batches ago
        if pipelined:
            # batch i's main kernels go to pipe j WITHOUT any host synchronisation, then the tail and the hand-over of batch i - 1 (other pipe)
            if args.long_reads:
                lb = long_batches[i % nb]
                res = db.query_device(lb["seq"].data_ptr(), lb["qinfo"].data_ptr(), nloc, lb["nchars"], max_win_ptr=lb["maxwin"].data_ptr(), second_pipe=bool(j),
                                      defer_tail=True)
            else:
                res = db.query_device(b.data_ptr(), qinfo.data_ptr(), nloc, nchars, max_win_uniform=max_win, second_pipe=bool(j), defer_tail=True)
            pend[j] = res.cands
            finish_pipe(j ^ 1)
            return res
        if mode == "K":
            res = db.query_device(b.data_ptr(), qinfo.data_ptr(), nloc, nchars, max_win_uniform=max_win, want_partial_hits=not numbers_wire,
                                  want_partial_numbers=numbers_wire)
            out_bufs[j].copy_(classify_key_sharded_device(db, res, nloc, K, max_win, wire=args.wire))   # all-to-all of the partial lists, rows 8-10 on the owner
            torch.cuda.current_stream().synchronize()
        elif args.long_reads:
            lb = long_batches[i % nb]
            res = db.query_device(lb["seq"].data_ptr(), lb["qinfo"].data_ptr(), nloc, lb["nchars"], max_win_ptr=lb["maxwin"].data_ptr())
            db.copy_results(out_bufs[j].data_ptr(), res.cands, nloc * K * 16)
            db.synchronize()
        else:
            res = db.query_device(b.data_ptr(), qinfo.data_ptr(), nloc, nchars, max_win_uniform=max_win)
            db.copy_results(out_bufs[j].data_ptr(), res.cands, nloc * K * 16)
            db.synchronize()
        if mode == "P":
            c = out_bufs[j]
            live = c[:, :, 1] > 0                            # part-local target numbers -> numbers in the whole collection
            c[:, :, 0] = torch.where(live, part_sel_t[c[:, :, 0].clamp(min=0, max=part_sel_t.numel() - 1).long()], c[:, :, 0])
            out_bufs[j].copy_(classify_partitioned(c))       # all-gather + merge: every rank holds the merged lists
            torch.cuda.current_stream().synchronize()
        elif dist_path:
            works[j] = gather_candidates_async(out_bufs[j], recv[j] if recv is not None else None, dst=0)
        return res

    def drain():
        if pipelined:
            finish_pipe(0); finish_pipe(1)
            db.synchronize()
        for j in range(nbuf):
            finish(j)

    for i in range(args.warmup):
        step(i)
    drain()
    torch.cuda.synchronize()
    first_batch[0] = args.warmup
    db.timing(True)
    db.timing_reset()

    def timed_region():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i)
        drain()                                              # every batch finished, every gather has arrived on rank 0
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    elapsed = timed_region()                                 # THE timed region: exactly K steps, max over ranks -> value
    db.timing(False)
    kt_timed = {k: db.timing_get(k) for k in KERNELS + (KERNELS_MODE_K if mode == "K" else ())} if rank == 0 else {}
    repeats = [elapsed] + [timed_region() for _ in range(max(0, args.repeats - 1))]   # the same K steps again: the spread (value_range)
    # per-kernel durations WITHOUT another batch's kernels beside them: a few steps one batch at a time (two batches in flight share the
    # device, the HIP events of the timed region bracket that sharing too)
    kt_solo = None
    if pipelined and rank == 0:
        db.timing(True); db.timing_reset()
        for i in range(min(3, args.steps)):
            if args.long_reads:
                lb = long_batches[i % nb]
                r = db.query_device(lb["seq"].data_ptr(), lb["qinfo"].data_ptr(), nloc, lb["nchars"], max_win_ptr=lb["maxwin"].data_ptr())
            else:
                r = db.query_device(batches[i % nb].data_ptr(), qinfo.data_ptr(), nloc, nchars, max_win_uniform=max_win)
            db.copy_results(out_bufs[0].data_ptr(), r.cands, nloc * K * 16)
            db.synchronize()
        db.timing(False)
        kt_solo = {k: db.timing_get(k) for k in KERNELS}
    host_fed = None
    if pipelined and world == 1 and args.host_fed and not args.force_dist:
        host_fed = host_fed_leg(db, batches if not args.long_reads else None, long_batches if args.long_reads else None, qinfo, nloc, nchars, max_win, K, args.steps, args.warmup, nb,
                                2 if pairs else 1, elapsed / args.steps)
    kstats = None
    if mode == "K" and rank == 0 and world == 1:
        # (the owner-side call resets the shard side's statistics: one more lookup pass for F and H of the line)
        db.query_device(batches[0].data_ptr(), qinfo.data_ptr(), nloc, nchars, max_win_uniform=max_win, want_partial_hits=not numbers_wire, want_partial_numbers=numbers_wire)
        db.synchronize()
        kstats = db.last_batch_stats()
    if world > 1:
        dist.barrier()

    if rank == 0:
        kt = kt_timed                                         # HIP events over THE timed region
        st = kstats or db.last_batch_stats()                  # of the last batch that went through the first pipe
        layout = db.table_layout()
        if cfg != 1:
            V = layout["location_bytes"]                       # SURVEY's V = bytes per location as the table holds them: 4 with the compact store
        per_read = 2 if pairs else 1                           # a pair counts as 2 reads (printing.cpp:605-608)
        F, H = st["features"] / (nloc * per_read), st["locations"] / (nloc * per_read)
        bytes_per_read = algorithmic_bytes_per_read(F, H, K, V)
        if args.long_reads:                                    # SURVEY's formula with the reads' own lengths: ceil(L/4) + ceil(L/8) summed over the last timed batch
            lb = long_batches[(args.warmup + args.steps - 1) % nb]
            bytes_per_read = float(((lb["lens"] + 3) // 4 + (lb["lens"] + 7) // 8).sum()) / nloc + 12.0 * F + V * H + 16.0 * K
        # the dominant KERNEL: timers that bracket several kernels (the sort's instances and merge passes, compaction + ordering, scans) are not candidates
        dom = max((k for k in kt if k not in MULTI_KERNEL_TIMERS), key=lambda k: kt[k][0])      # (mode K: its own kernels are candidates too)
        dom_ms = kt[dom][0] / max(kt[dom][1], 1)
        achieved = bytes_per_read * nloc * per_read / (dom_ms * 1e-3) / 1e9
        L_mean = float(long_batches[(args.warmup + args.steps - 1) % nb]["lens"].mean()) if args.long_reads else float(READ_LEN)
        kshare = kernel_bytes_per_read(dom, L_mean, F, H, K, V)
        kbytes = None if kshare is None else kshare * nloc * per_read
        if cfg == 2 and layout.get("list_align", 1) <= 1 and layout["location_bytes"] == 4:
            pmc_tag += "_plain"                                # (the passes with MC_LIST_ALIGN=0: lists at arbitrary 4-byte offsets)
        traffic, traffic_src, traffic_detail = measured_traffic(dom, pmc_tag)
        if traffic is not None:
            same, why = pmc_layout_matches(pmc_tag, layout)
            if not same:                                       # a request count of another table layout is not this run's traffic
                traffic, traffic_src, traffic_detail = None, None, {"refused": why}
        _, pmc_kernels = _pmc_rows(dom, pmc_tag)
        total_reads = world * args.steps * B * per_read
        value = total_reads / elapsed * 60.0 / 1e6
        result = {
            "metric": "Mreads/min (long reads: 200-19000 bp, median 480)" if args.long_reads else "Mreads/min (150 bp)",
            "value": round(value, 2), "unit": "Mreads/min", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": workload, "reads_per_step_per_gpu": B,
                       "maxcand": K, "k": db.k, "sketchlen": db.s, "winlen": db.w, "winstride": db.stride,
                       "db_targets": db_info[5], "db_locations": db_info[7], "db_build_s": round(build_s, 2), "db_build": build_info, "load_factor": lf,
                       "hbm_used_GB": round((torch.cuda.mem_get_info()[1] - torch.cuda.mem_get_info()[0]) / 1e9, 1),
                       "mode": mode, "pairs": pairs,
                       "parallelism": {"R": f"replicated DB x{world}, reads sharded, RCCL gather of top candidates",
                                       "P": f"{world} parts, one per GPU; all reads against every part, RCCL all-gather of per-part candidates, merge",
                                       "K": f"1 part key-sharded over {world} GPUs; every shard looks up the features it owns for all reads, RCCL all-to-all-v of the partial "
                                            f"location lists ({args.wire} bytes per location), candidates on the read's owner, gather"}[mode]},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6),
                         # the same bytes over the WHOLE step (all kernels, launches, the copy of the candidates)
                         "step_frac": round(bytes_per_read * nloc * per_read / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 6),
                         "traffic": traffic, "traffic_source": traffic_src, "traffic_detail": traffic_detail,
                         # like for like with `traffic`: the dominant kernel's OWN share of the algorithmic bytes (kernel_bytes_per_read)
                         # the kernel's whole name as the rocprofv3 summaries carry it (template arguments and all), where a summary of this workload exists
                         "kernel_name": "mcamd::" + (sorted(pmc_kernels)[0] if pmc_kernels else KERNEL_FULL.get(dom, KERNEL_OF.get(dom, (dom,))[0])),
                         "kernel_algorithmic_bytes": None if kbytes is None else round(kbytes),
                         "kernel_frac": None if kbytes is None else round(kbytes / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6),
                         "traffic_over_kernel_bytes": None if (kbytes is None or traffic is None) else round(traffic / kbytes, 3),
                         "algorithmic_bytes_per_launch": round(bytes_per_read * nloc * per_read),
                         "bytes_per_read": round(bytes_per_read, 1), "F": round(F, 3), "H": round(H, 3), "V": V,
                         "table_location_bytes": layout["location_bytes"], "table_list_align": layout.get("list_align", 1), "table_list_entries": layout["list_locations"],
                         "table_direct_index": bool(layout.get("direct_index", False)),
                         "kernel_ms": {k: round(v[0] / max(v[1], 1), 4) for k, v in kt.items()}},
        }
        vals = [total_reads / e * 60.0 / 1e6 for e in repeats]
        result["value_range"] = [round(min(vals), 1), round(max(vals), 1)]
        result["repeats_ms_per_step"] = [round(e / args.steps * 1e3, 3) for e in repeats]
        result["config"]["batches_in_flight"] = 2 if pipelined else 1
        if host_fed is not None:
            result["host_fed"] = host_fed
        if kt_solo is not None:
            # two batches in flight share the device: the events of the timed region bracket that sharing.  The same kernels one batch at a
            # time (3 steps after the timed region):
            sm = {k: round(v[0] / max(v[1], 1), 4) for k, v in kt_solo.items()}
            sdom = max((k for k in sm if k not in MULTI_KERNEL_TIMERS), key=lambda k: sm[k])
            result["roofline"]["kernel_ms_solo"] = sm
            result["roofline"]["kernel_solo"] = sdom
            result["roofline"]["frac_solo"] = round(bytes_per_read * nloc * per_read / (sm[sdom] * 1e-3) / 1e9 / HBM_PEAK_GBS, 6)
            result["roofline"]["step_ms_solo"] = round(sum(sm.values()), 3)
        # second roofline: 64-byte read requests per second of the dominant kernel against the box's measured random-access
        # peak for that kernel's access shape.  Requests per launch: TCC_EA0_RDREQ of the committed PMC pass (same batch size only).
        req = measured_requests(dom, pmc_tag) if pmc_layout_matches(pmc_tag, layout)[0] else None
        shape = "wave_512B_list" if dom.startswith(("big_", "gw_", "hash_", "mid_")) else \
                ("quad_64B" if db_info[7] > 2_000_000_000 else "lane_private_64B")
        if peak is not None:
            ra = {"shape": shape, "peak_requests_per_s": round(peak[shape]), "peak_all_shapes": {k: round(v) if k != "buffer_GiB" else v for k, v in peak.items()}}
            if req is not None:
                ra["requests_per_launch"] = req
                ra["requests_per_s"] = round(req / (dom_ms * 1e-3))
                ra["frac"] = round(ra["requests_per_s"] / max(peak[shape], 1.0), 4)
            else:
                ra["requests_per_s"] = None; ra["frac"] = None
            result["roofline"]["random_access"] = ra
        if args.long_reads:
            tot_bases = sum(long_batches[(args.warmup + i) % nb]["bases"] for i in range(args.steps))
            result["config"]["Gbases_per_s"] = round(world * tot_bases / elapsed / 1e9, 3)
            result["config"]["mean_read_len"] = round(tot_bases / (args.steps * B), 1)
        first_batch[0] = 0                                    # the checker legs look at batch 0
        if world == 1 and args.cpu_seconds > 0 and args.long_reads:
            cb, par = cpu_leg_long_reads(spec, db, long_batches[0], K, args.parity_reads, args.cpu_seconds, lambda: (step(0), drain(), db.synchronize()), out_cands)
            result["cpu_baseline"] = cb
            result["parity"] = par
        elif world == 1 and args.cpu_seconds > 0:
            step(0)
            drain()
            db.synchronize()
            n_chk = B if cfg == 1 else min(B, args.parity_reads if not pairs else min(args.parity_reads, 20_000))
            gpu_c = out_cands[:n_chk].cpu().numpy().view(np.uint32).reshape(n_chk, K, 4)
            gc = np.zeros((n_chk, K), dtype=api.cand_dtype)
            gc["tgt"], gc["hits"], gc["beg"], gc["end"] = gpu_c[..., 0], gpu_c[..., 1], gpu_c[..., 2], gpu_c[..., 3]
            reads_host = batches[0][: n_chk * PAD_LEN].reshape(n_chk, PAD_LEN).cpu().numpy()
            mates_host = batches[0][nloc * PAD_LEN: (nloc + n_chk) * PAD_LEN].reshape(n_chk, PAD_LEN).cpu().numpy() if pairs else None
            if cfg == 1:
                cb, par = cpu_leg_config1(os.path.join(dbdir, "syn16"), reads_host, gc, K, args.cpu_seconds)
            elif args.reference_files and mode == "R" and not pairs:
                cb, par = cpu_leg_reference_files(args.reference_files, reads_host, gc, K, args.cpu_seconds)
            else:
                cb, par = cpu_leg_config2(spec, reads_host, gc, K, n_chk, args.cpu_seconds, mates_host)
            result["cpu_baseline"] = cb
            result["parity"] = par
    db.close()
    if rank == 0 and world == 1 and cfg == 2 and mode == "R" and not pairs and not args.long_reads and args.cpu_seconds > 0 and args.calibrate_scale > 0 \
            and "cpu_baseline" in result and not args.reference_files:
        torch.cuda.empty_cache()
        cal = reference_calibration(args.calibrate_scale, K, lf, local, args.cpu_seconds)
        if cal is None and args.calibrate_scale > 0.05:         # the files of that cut do not fit this box's memory allowance: the smaller cut
            cal = reference_calibration(0.05, K, lf, local, args.cpu_seconds)
        if cal:
            result["cpu_baseline"]["reference_calibration"] = cal
    # ---- the sharded forms of the path on THESE N GPUs, checked against the oracle (outside every timing): its own job, rank 0 starts it
    # and the other ranks wait for it at the rendezvous store -- on the host, their GPUs are free for the job's ranks
    if cfg == 2 and mode == "R" and args.selfcheck_seconds > 0 and (world > 1 or args.cpu_seconds > 0):
        batches = out_bufs = None
        torch.cuda.empty_cache()
        store = dist.distributed_c10d._get_default_store() if dist.is_initialized() else None
        if rank == 0:
            result["multi_gpu_selfcheck"] = run_multi_gpu_selfcheck(world, args.selfcheck_seconds)
            if store is not None:
                store.set("mc_selfcheck_done", "1")
        elif store is not None:
            from datetime import timedelta
            try:
                store.wait(["mc_selfcheck_done"], timedelta(seconds=args.selfcheck_seconds + 120))
            except Exception:                                   # noqa: BLE001
                pass
    if rank == 0 and world == 1 and cfg == 2 and mode == "R" and not pairs and not args.long_reads and args.scale == 1.0 and args.shape == "default" \
            and args.cpu_seconds > 0 and args.e2e_scale > 0:
        batches = out_bufs = mates = None                       # (the device belongs to the mcq processes now)
        torch.cuda.empty_cache()
        try:                                                    # ... and the host: the host-fed leg's pinned batches (15 GB) sit in torch's pinned-memory
            torch._C._host_emptyCache()                         # cache; beside a 174 GB file set in /dev/shm they pushed the second mcq run's load from 9 to 68 s
        except Exception:                                       # noqa: BLE001
            pass
        result["e2e"] = e2e_leg(args.e2e_scale, args.e2e_seconds)
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
