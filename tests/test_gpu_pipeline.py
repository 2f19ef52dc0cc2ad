"""GPU: two batches in flight from one caller thread -- mc_query_device(MC_DEFER_TAIL) on alternating pipes, mc_query_finish after the
next batch has been enqueued (what bench.py's timed region does) -- must leave exactly the candidates the one-batch-at-a-time call
leaves, also for the reads whose work IS the deferred tail: long reads (sorted lists), reads the filtered path hands back to the exact
wave kernel, reads with duplicate hashes.  And the part groups in the caller's hands (mc_partset_select_group /
mc_partset_classify_resident, what `mcq -resident-parts` streams through) against mc_partset_classify.
Reference for the overlap: query_batch.cuh:369-371 (one stream per batch), database_query.hpp:110-113."""
import numpy as np
import pytest

import cpuref
import scale_util
from metacache_amd import api, synth, synthdb

pytestmark = pytest.mark.gpu
THREADS = 8


def _device_batch(reads, dev):
    import torch
    pad = [len(r) + (-len(r)) % 4 for r in reads]
    offs = np.concatenate([[0], np.cumsum(pad)]).astype(np.int64)
    buf = np.zeros(int(offs[-1]) + 16, dtype=np.uint8)
    for r, o in zip(reads, offs[:-1]):
        buf[o:o + len(r)] = np.frombuffer(r, dtype=np.uint8)
    qinfo = np.zeros((len(reads), 4), dtype=np.uint32)
    qinfo[:, 0] = offs[:-1]; qinfo[:, 1] = [len(r) for r in reads]; qinfo[:, 2] = offs[:-1]
    mw = np.array([2 + len(r) // 112 for r in reads], dtype=np.int32)
    return (torch.from_numpy(buf).to(dev), torch.from_numpy(qinfo.view(np.int32)).to(dev), torch.from_numpy(mw).to(dev), int(offs[-1]))


@pytest.mark.parametrize("K,lowest", [(2, 0), (3, 4)])
def test_deferred_tail_on_two_pipes_equals_the_synchronous_call(monkeypatch, K, lowest):
    import torch
    monkeypatch.setenv("MC_BIG_MIN", "0")                      # every list above 64 locations takes the filtered path
    spec = synthdb.phylogeny(4, 2, 10, 30_000, 40_000, seed=4242, div_strain=(0.002, 0.01))
    db, _ = synthdb.build_database(spec, shards=1, max_candidates=K)
    assert db.table_layout()["location_bytes"] == 4
    db.set_lineages(spec.lineages())
    cs = synthdb.CpuSynth()
    rng = np.random.default_rng(5)
    batches = []
    for b, lens in enumerate(((150,), (150, 300, 512), (700, 1500, 4000, 150), (150, 9000))):
        reads = []
        for j, L in enumerate(lens):
            P = synthdb.read_params(spec, 2000 + 10 * b + j, read_len=L, sub_rate=0.03 if L > 600 else 0.01)
            reads += [bytes(r[:L]) for r in cs.reads(spec, P, 0, 400 if L <= 512 else 60)]
        reads += [bytes(synth.random_genome(rng, 300)) for _ in range(20)]      # nothing found: empty lists
        reads += [b"ACGT" * 60, b"A" * 200]                                     # duplicate hashes inside a sketch: the exact wave kernel
        batches.append([reads[i] for i in rng.permutation(len(reads))])
    dev = torch.device("cuda", 0)
    dbatch = [_device_batch(r, dev) for r in batches]
    torch.cuda.synchronize()
    # one batch at a time
    want = []
    for (seq, qi, mw, nch), reads in zip(dbatch, batches):
        r = db.query_device(seq.data_ptr(), qi.data_ptr(), len(reads), nch, max_win_ptr=mw.data_ptr(), lowest=lowest)
        out = torch.empty((len(reads), K, 4), dtype=torch.int32, device=dev)   # (empty: a fill kernel on torch's stream would race the copy on the context's)
        db.copy_results(out.data_ptr(), r.cands, len(reads) * K * 16); db.synchronize()
        want.append(out.cpu().numpy().view(np.uint32))
    # ... against the oracle (the long reads' sorted lists, the handed-back reads)
    odb = scale_util.oracle_database(spec, None, threads=THREADS, with_lineages=True)
    for reads, w in zip(batches, want):
        for i in range(0, len(reads), 7):
            _, e = odb.query(reads[i], b"", K, lowest, 0)
            for j in range(K):
                exp = (int(e[j]["tgt"]), int(e[j]["hits"]), int(e[j]["beg"]), int(e[j]["end"])) if j < len(e) else None
                got = tuple(int(x) for x in w[i, j]) if w[i, j, 1] else None
                assert got == exp, (i, j, len(reads[i]), w[i], e)
    odb.close()
    # two in flight: batch i on pipe i & 1 with the tail deferred, the tail of batch i - 1 after batch i has been enqueued
    got = [None] * len(batches)
    pend = {}

    def finish(j):
        if j in pend:
            i, ptr, n = pend.pop(j)
            db.query_finish(second_pipe=bool(j))
            out = torch.empty((n, K, 4), dtype=torch.int32, device=dev)
            db.copy_results(out.data_ptr(), ptr, n * K * 16, second_pipe=bool(j))
            db.query_wait(second_pipe=bool(j))
            got[i] = out.cpu().numpy().view(np.uint32)
    for rep in range(2):                                       # the second round reuses both pipes' workspaces
        for i, ((seq, qi, mw, nch), reads) in enumerate(zip(dbatch, batches)):
            j = i & 1
            finish(j)
            r = db.query_device(seq.data_ptr(), qi.data_ptr(), len(reads), nch, max_win_ptr=mw.data_ptr(), lowest=lowest, second_pipe=bool(j), defer_tail=True)
            pend[j] = (i, r.cands, len(reads))
            finish(j ^ 1)
        finish(0); finish(1)
        db.synchronize()
        for i in range(len(batches)):
            live = (got[i][:, :, 1] > 0) | (want[i][:, :, 1] > 0)
            bad = ((got[i] != want[i]).any(axis=2) & live).any(axis=1)
            assert not bad.any(), (rep, i, int(bad.sum()), int(np.flatnonzero(bad)[0]))
    # a pending tail that nobody asks for is run by the next call on that pipe; results of the new call are complete
    seq, qi, mw, nch = dbatch[2]
    db.query_device(seq.data_ptr(), qi.data_ptr(), len(batches[2]), nch, max_win_ptr=mw.data_ptr(), lowest=lowest, defer_tail=True)
    r = db.query_device(seq.data_ptr(), qi.data_ptr(), len(batches[2]), nch, max_win_ptr=mw.data_ptr(), lowest=lowest)
    out = torch.empty((len(batches[2]), K, 4), dtype=torch.int32, device=dev)
    db.copy_results(out.data_ptr(), r.cands, len(batches[2]) * K * 16); db.synchronize()
    o = out.cpu().numpy().view(np.uint32)
    live = (o[:, :, 1] > 0) | (want[2][:, :, 1] > 0)
    assert not ((o != want[2]).any(axis=2) & live).any()
    db.close()


@pytest.mark.parametrize("resident,lowest", [(1, 0), (2, 4), (3, 0)])
def test_part_groups_streamed_batch_by_batch_equal_classify(golden, resident, lowest, monkeypatch):
    monkeypatch.setenv("MC_PARTSET_RCCL", "1")
    single, p1, p2 = golden.reads()
    name = golden.db_path("toy32p4")
    K = 2
    ps = api.PartSet(name, resident=resident, max_candidates=K, slot_max_queries=400, slot_max_chars=1 << 17)
    p1, p2 = p1[:600], p2[:600]
    npairs = len(p1)
    want = ps.classify(single[:1500], lowest=lowest)
    wantp = ps.classify(p1, p2, lowest=lowest, insert_max=700)
    groups = ps.info()["groups"]
    # group by group, the reads in batches of their own; between the groups a batch keeps nothing but its candidate lists
    cuts = [(0, 500), (500, 1100), (1100, 1500)]
    got = np.zeros((1500, K), dtype=api.cand_dtype)
    gotp = np.zeros((npairs, K), dtype=api.cand_dtype)
    for g in range(groups):
        ps.select_group(g)
        for lo, hi in cuts:
            ps.classify_resident(single[lo:hi], None, got[lo:hi], has_prior=g > 0, lowest=lowest)
        ps.classify_resident(p1, p2, gotp, has_prior=g > 0, lowest=lowest, insert_max=700)
    info = ps.info()
    ps.close()
    for a, b in ((got, want), (gotp, wantp)):
        for f in ("hits", "beg", "end"):
            assert np.array_equal(a[f], b[f]), f
        used = b["hits"] > 0
        assert np.array_equal(a["tgt"][used], b["tgt"][used])
    assert info["load_bytes"] > 0 and info["groups"] == -(-4 // resident)
    # against the oracle's multi-part semantics as well
    odb = cpuref.oracle().open(name)
    for i in range(0, 1500, 11):
        _, c = odb.query(single[i], b"", K, lowest, 0, mode=1)
        for j in range(K):
            if j < len(c):
                assert (got[i][j]["tgt"], got[i][j]["hits"], got[i][j]["beg"], got[i][j]["end"]) == (c[j]["tgt"], c[j]["hits"], c[j]["beg"], c[j]["end"]), (i, j)
            else:
                assert got[i][j]["hits"] == 0
    odb.close()


@pytest.mark.parametrize("big_min", [None, 0])
def test_batch_sizes_around_the_small_batch_kernels_limits(golden, big_min):
    """Batches of up to 16 384 / 32 768 reads take other steps on the stream than larger ones (kernels.hip: scan_small_kernel, plan_scan_small_kernel,
    flag_count_small_kernel; context.cpp: one host round trip for the sorted class and the wave tail): a read's candidates must not depend on
    the size of the batch it came in.  big_min = 0 sends the toy table's lists through the filtered path and its counting kernels."""
    single, _, _ = golden.reads()
    rng = np.random.default_rng(4096)
    reads = [single[k] for k in rng.integers(0, len(single), 40000)]
    chars = sum((len(r) + 3) // 4 * 4 for r in reads) + 64

    def run(n_batch, n_reads):
        db = api.Database.open(golden.db_path("toy32"), max_candidates=2, copy_allhits=0, slot_max_queries=n_batch, slot_max_chars=chars)
        try:
            if big_min is not None:
                db.set_tuning("big_min", big_min)
            c, _, _ = db.query(reads[:n_reads], lowest=0)
            return c
        finally:
            db.close()

    whole = run(40000, 40000)                                    # one batch beyond every limit
    for n in (1, 4095, 4096, 4097, 16383, 16384, 16385, 32767, 32768, 32769):
        got = run(n, n)
        assert np.array_equal(got, whole[:n]), n
    # ... nor on the batches before it on the same pipe (sizes going up and down: the pinned counters and the workspaces are reused)
    db = api.Database.open(golden.db_path("toy32"), max_candidates=2, copy_allhits=0, slot_max_queries=5000, slot_max_chars=chars)
    try:
        if big_min is not None:
            db.set_tuning("big_min", big_min)
        c, _, _ = db.query(reads[:23000], lowest=0)              # 5 000, 5 000, 5 000, 5 000, 3 000
        assert np.array_equal(c, whole[:23000])
    finally:
        db.close()
