"""CPU, world_size 2, gloo: the N>1 driver logic (read sharding + gather of per-rank candidate lists).
The per-rank compute is stood in for by the C oracle (the HIP path needs a GPU); what is under test is
metacache_amd/distributed.py, which bench.py and multi-GPU callers use unchanged with RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from metacache_amd.distributed import shard_bounds


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n, K, q):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here); sys.path.insert(0, os.path.dirname(here))
    import cpuref
    from metacache_amd.distributed import classify_sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    gold = os.path.join(here, "golden")
    z = np.load(os.path.join(gold, "toy_reads.npz"))
    off = z["single_off"]
    reads = [z["single"][int(off[i]):int(off[i + 1])].tobytes() for i in range(n)]
    db = cpuref.oracle().open(os.path.join(gold, "toy32"))

    def classify(lo, hi):
        out = torch.zeros((hi - lo, K, 4), dtype=torch.int32)
        for i in range(lo, hi):
            _, c = db.query(reads[i], b"", K, 0, 0)
            for j in range(len(c)):
                out[i - lo, j] = torch.tensor([int(c[j]["tgt"]), int(c[j]["hits"]), int(c[j]["beg"]), int(c[j]["end"])], dtype=torch.int64).to(torch.int32)
        return out

    res = classify_sharded(n, classify)
    if rank == 0:
        q.put(res.numpy())
    else:
        assert res is None
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 100, 101):
        for w in (1, 2, 3, 8):
            b = [shard_bounds(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1


def test_two_ranks_gloo_gather_matches_single_process(golden):
    import cpuref
    n, K, world = 101, 2, 2          # odd count: shards of different size
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, K, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    single, _, _ = golden.reads()
    db = cpuref.oracle().open(golden.db_path("toy32"))
    assert got.shape == (n, K, 4)
    for i in range(n):
        _, c = db.query(single[i], b"", K, 0, 0)
        for j in range(K):
            if j < len(c):
                assert [int(x) for x in got[i, j].view(np.uint32)] == [int(c[j]["tgt"]), int(c[j]["hits"]), int(c[j]["beg"]), int(c[j]["end"])]
            else:
                assert got[i, j, 1] == 0
    db.close()


def _worker_parts(rank, world, port, n, K, q):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here); sys.path.insert(0, os.path.dirname(here))
    import cpuref
    from metacache_amd.distributed import classify_partitioned
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    gold = os.path.join(here, "golden")
    z = np.load(os.path.join(gold, "toy_reads.npz"))
    off = z["single_off"]
    reads = [z["single"][int(off[i]):int(off[i + 1])].tobytes() for i in range(n)]
    db = cpuref.oracle().open_part(os.path.join(gold, "toy32p2"), rank)          # this rank's part only
    local = torch.zeros((n, K, 4), dtype=torch.int32)
    for i, r in enumerate(reads):
        _, c = db.query(r, b"", K, 0, 0)
        for j in range(len(c)):
            local[i, j] = torch.tensor([int(c[j]["tgt"]), int(c[j]["hits"]), int(c[j]["beg"]), int(c[j]["end"])], dtype=torch.int64).to(torch.int32)
        for j in range(len(c), K):
            local[i, j, 0] = -1
    merged = classify_partitioned(local)
    if rank == 0:
        q.put(merged.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_mode_p_one_part_per_rank_matches_intended_multipart(golden):
    """Mode P: rank r holds part r of the 2-part reference DB; merged per-part top-K == the oracle's intended
    multi-part result on the whole database."""
    import cpuref
    n, K, world = 300, 2, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_parts, args=(r, world, port, n, K, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    single, _, _ = golden.reads()
    db = cpuref.oracle().open(golden.db_path("toy32p2"))
    for i in range(n):
        _, c = db.query(single[i], b"", K, 0, 0, mode=1)
        for j in range(K):
            if j < len(c):
                assert [int(x) for x in got[i, j].view(np.uint32)] == [int(c[j]["tgt"]), int(c[j]["hits"]), int(c[j]["beg"]), int(c[j]["end"])], (i, j)
            else:
                assert got[i, j, 1] == 0
    db.close()


def _worker_ranges(rank, world, port, n, K, lowest, q):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here); sys.path.insert(0, os.path.dirname(here))
    import cpuref
    from metacache_amd.distributed import classify_partitioned
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    gold = os.path.join(here, "golden")
    z = np.load(os.path.join(gold, "toy_reads.npz"))
    off = z["single_off"]
    reads = [z["single"][int(off[i]):int(off[i + 1])].tobytes() for i in range(n)]
    db = cpuref.oracle().open(os.path.join(gold, "toy32"))
    nt = db.n_targets
    lo, hi = rank * nt // world, (rank + 1) * nt // world      # this rank's contiguous target range
    local = torch.zeros((n, K, 4), dtype=torch.int32)
    local[:, :, 0] = -1
    for i, r in enumerate(reads):
        # what a context holding only the range's locations answers: the whole table's candidates of the range's targets, in their order
        # (a candidate never spans two targets, candidate_generation.hpp:96-150), the first K of them
        _, c = db.query(r, b"", 64, lowest, 0)
        mine = [x for x in c if lo <= int(x["tgt"]) < hi][:K]
        for j, x in enumerate(mine):
            local[i, j] = torch.tensor([int(x["tgt"]), int(x["hits"]), int(x["beg"]), int(x["end"])], dtype=torch.int64).to(torch.int32)
    merged = classify_partitioned(local)                       # all-gather + merge in rank order = range order
    if rank == 0:
        q.put(merged.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_mode_t_contiguous_target_ranges_merge_to_the_whole_table(golden):
    """Mode T (mc_config.target_shard_*; the GPU side: tests/test_gpu_target_ranges.py): rank r answers for a contiguous range of the
    targets; the ranks' top lists merged in rank order are the whole table's top list (sequence level: the merge this mode rests on)."""
    import cpuref
    n, K, world = 300, 2, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_ranges, args=(r, world, port, n, K, 0, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    single, _, _ = golden.reads()
    db = cpuref.oracle().open(golden.db_path("toy32"))
    for i in range(n):
        _, c = db.query(single[i], b"", K, 0, 0)
        for j in range(K):
            if j < len(c):
                assert [int(x) for x in got[i, j].view(np.uint32)] == [int(c[j]["tgt"]), int(c[j]["hits"]), int(c[j]["beg"]), int(c[j]["end"])], (i, j)
            else:
                assert got[i, j, 1] == 0
    db.close()


# ---- Mode K: features key-sharded over the ranks, partial location lists exchanged to the read's owner -------------------
def _worker_mode_k(rank, world, port, n, K, q):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here); sys.path.insert(0, os.path.dirname(here))
    import cpuref
    from metacache_amd import api
    from metacache_amd.distributed import classify_key_sharded, gather_candidates
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    gold = os.path.join(here, "golden")
    z = np.load(os.path.join(gold, "toy_reads.npz"))
    off = z["single_off"]
    reads = [z["single"][int(off[i]):int(off[i + 1])].tobytes() for i in range(n)]
    orc = cpuref.oracle()
    db = orc.open(os.path.join(gold, "toy32"))
    owner = api.lib().mc_key_owner
    # step 1 (stand-in for the sharded GPU context): this rank's partial lists = lookups of the features it owns
    counts, chunks = [], []
    for r in reads:
        feats, cnt = orc.sketch(r, db.k, db.s, db.w, db.stride)
        mine = [db.lookup(int(f)) for w in range(len(cnt)) for f in feats[w, :cnt[w]] if owner(int(f), world) == rank]
        lst = np.concatenate(mine) if mine else np.zeros(0, dtype=np.uint64)
        counts.append(len(lst)); chunks.append(lst)
    counts_t = torch.tensor(counts, dtype=torch.int64)
    hits_t = torch.from_numpy(np.concatenate(chunks).astype(np.int64)) if sum(counts) else torch.zeros(0, dtype=torch.int64)

    def candidates(offsets, union):
        m = offsets.numel() - 1
        lo, _ = shard_bounds(n, rank, world)
        out = torch.zeros((m, K, 4), dtype=torch.int32)
        u = union.numpy().astype(np.uint64)
        for i in range(m):
            lst = np.sort(u[int(offsets[i]):int(offsets[i + 1])])                # row 8 on the union
            max_win = 2 + len(reads[lo + i]) // db.stride
            c = orc.candidates(lst, max_win, K)
            for j in range(min(K, len(c))):
                out[i, j] = torch.tensor([int(c[j]["tgt"]), int(c[j]["hits"]), int(c[j]["beg"]), int(c[j]["end"])], dtype=torch.int64).to(torch.int32)
        return out

    # the exchange of the device data path (one size round trip, int32 counts source-major) must deliver what the reference form does
    from metacache_amd.distributed import exchange_partial_hits, exchange_partial_lists
    offs_t = torch.zeros(n + 1, dtype=torch.int64); offs_t[1:] = torch.cumsum(counts_t, 0)
    psc, psh = exchange_partial_hits(counts_t, hits_t)
    cnt32, rh, total = exchange_partial_lists(offs_t, hits_t)
    assert torch.equal(cnt32.view(world, -1).to(torch.int64), psc) and total == sum(int(x.numel()) for x in psh)
    assert torch.equal(rh, torch.cat(psh) if total else rh)
    # ... and so must the exchange of the 4-byte wire (numbers instead of (target, window) pairs; split sizes from the host-side cuts)
    from metacache_amd.distributed import exchange_numbers
    num_t = (hits_t & 0x7FFFFFFF).to(torch.int32)
    bnds = [shard_bounds(n, r, world) for r in range(world)]
    cuts = [int(offs_t[b[0]]) for b in bnds] + [int(offs_t[n])]
    rc4, rn4, so4 = exchange_numbers(counts_t.to(torch.int32), num_t, cuts, n)
    assert torch.equal(rc4, cnt32) and so4[-1] == total and torch.equal(rn4[:total], (rh & 0x7FFFFFFF).to(torch.int32))
    assert [so4[i + 1] - so4[i] for i in range(world)] == [int(x.numel()) for x in psh]
    local = classify_key_sharded(counts_t, hits_t, candidates)
    parts = gather_candidates(local, dst=0)
    if rank == 0:
        q.put(torch.cat(parts, dim=0).numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_gloo_mode_k_union_equals_whole_table(golden):
    import cpuref
    n, K, world = 160, 2, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_mode_k, args=(r, world, port, n, K, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    single, _, _ = golden.reads()
    db = cpuref.oracle().open(golden.db_path("toy32"))
    some = 0
    for i in range(n):
        _, c = db.query(single[i], b"", K, 0, 0)
        for j in range(K):
            if j < len(c):
                assert tuple(int(x) for x in got[i, j].view(np.uint32)) == (int(c[j]["tgt"]), int(c[j]["hits"]), int(c[j]["beg"]), int(c[j]["end"])), (i, j)
                some += 1
            else:
                assert got[i, j, 1] == 0
    assert some > n
    db.close()


def _worker_async_gather(rank, world, port, q):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here))
    from metacache_amd.distributed import gather_candidates_async
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m, K = 50, 2
    bufs = [torch.zeros((m, K, 4), dtype=torch.int32) for _ in range(2)]
    recv = [[torch.zeros((m, K, 4), dtype=torch.int32) for _ in range(world)] for _ in range(2)] if rank == 0 else None
    works = [None, None]
    seen = []
    for step in range(5):                                   # double buffering as in bench.py
        j = step % 2
        if works[j] is not None:
            works[j].wait()
            if rank == 0:
                seen.append(torch.stack(recv[j]).clone())
        bufs[j].fill_(1000 * step + rank)
        works[j] = gather_candidates_async(bufs[j], recv[j] if recv is not None else None, dst=0)
    for j in ((5 % 2), (6 % 2)):
        if works[j] is not None:
            works[j].wait()
            if rank == 0:
                seen.append(torch.stack(recv[j]).clone())
    if rank == 0:
        q.put(torch.stack(seen).numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_gloo_async_gather_double_buffered():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_async_gather, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)                                # [5 steps, world, m, K, 4]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got.shape[0] == 5
    for step in range(5):
        for r in range(world):
            assert (got[step, r] == 1000 * step + r).all()


@pytest.mark.parametrize("lowest,K", [(4, 1), (4, 2), (4, 3), (6, 2), (6, 4)])
def test_mode_p_taxon_merging_equals_intended_multipart(golden, lowest, K):
    """Mode P with -lowest above sequence: the merge of the per-part top-K lists (each merged per taxon inside its part) replays the
    reference's per-taxon insert (candidate_generation.hpp:203-228) over the parts -- against the oracle's intended multi-part
    semantics on the whole 2-part database (no process group needed: the merge itself is what is new)."""
    import cpuref
    from metacache_amd.distributed import merge_part_candidates
    orc = cpuref.oracle()
    single, p1, p2 = golden.reads()
    reads = [(s, b"") for s in single[:700]] + list(zip(p1[:200], p2[:200]))
    n = len(reads)
    whole = orc.open(golden.db_path("toy32p2"))
    parts = [orc.open_part(golden.db_path("toy32p2"), p) for p in range(2)]
    per_part, per_tax = [], []
    for pdb in parts:
        c4 = torch.zeros((n, K, 4), dtype=torch.int32); c4[:, :, 0] = -1
        tx = torch.zeros((n, K), dtype=torch.int64)
        for i, (a, b) in enumerate(reads):
            _, c = pdb.query(a, b, K, lowest, 0)
            for j in range(min(K, len(c))):
                c4[i, j] = torch.tensor([int(c[j]["tgt"]), int(c[j]["hits"]), int(c[j]["beg"]), int(c[j]["end"])], dtype=torch.int64).to(torch.int32)
                tx[i, j] = int(c[j]["taxid"])
        per_part.append(c4); per_tax.append(tx)
    got = merge_part_candidates(per_part, per_tax).numpy()
    nontrivial = 0
    for i, (a, b) in enumerate(reads):
        _, e = whole.query(a, b, K, lowest, 0, mode=1)
        e = e[:K]
        for j in range(K):
            if j < len(e):
                assert [int(x) for x in got[i, j].view(np.uint32)] == [int(e[j]["tgt"]) & 0xFFFFFF, int(e[j]["hits"]), int(e[j]["beg"]), int(e[j]["end"])], (i, j, got[i], e)
            else:
                assert got[i, j, 1] == 0, (i, j, got[i], e)
        nontrivial += int(len(e) > 0 and any(int(per_part[1][i, j, 1]) > 0 for j in range(K)) and any(int(per_part[0][i, j, 1]) > 0 for j in range(K)))
    assert nontrivial > 100                                       # both parts contributed for many reads
    whole.close()
    for pdb in parts:
        pdb.close()
