"""GPU: what BASELINE.json configs[2] (RefSeq-scale table) stands on -- the synthetic collection on the device, the builder's
device / key-shard / streaming-table path, the filtered candidate kernel for location lists beyond 1024, and one database
large enough that the quad bucket fetch is chosen automatically -- all against the C oracle, which builds the buckets it needs
ITSELF from the same collection (oracle/mc_oracle.c: mco_db_build), bit-exact."""
import os

import numpy as np
import pytest

import cpuref
import scale_util
from metacache_amd import api, synth, synthdb

pytestmark = pytest.mark.gpu

THREADS = max(4, 2 * scale_util.effective_cpus())      # the GPU boxes show 256 CPUs and grant 16


def _check(got, e, K, tag):
    e = e[:K]
    for j in range(K):
        if j < len(e):
            assert (got[j]["tgt"], got[j]["hits"], got[j]["beg"], got[j]["end"]) == (e[j]["tgt"], e[j]["hits"], e[j]["beg"], e[j]["end"]), (tag, j, got, e)
        else:
            assert got[j]["hits"] == 0, (tag, j, got, e)


def test_synthetic_collection_gpu_equals_cpu():
    import torch
    spec = synthdb.phylogeny(3, 2, 3, 20_001, 40_003, seed=7)
    cs, gs = synthdb.CpuSynth(), synthdb.GpuSynth(0)
    off = spec.offsets(0, len(spec.targets))
    buf = torch.zeros(int(off[-1]) + 64, dtype=torch.uint8, device="cuda:0")
    gs.targets(spec, 0, len(spec.targets), buf)
    host = buf.cpu().numpy()
    for t in range(len(spec.targets)):
        L = int(spec.targets["length"][t])
        assert np.array_equal(host[int(off[t]):int(off[t]) + L], cs.target(spec, t)), t
    for paired in (False, True):
        P = synthdb.read_params(spec, 99, paired=paired)
        n = 5000
        a = torch.zeros((n, P.row_bytes), dtype=torch.uint8, device="cuda:0")
        b = torch.zeros((n, P.row_bytes), dtype=torch.uint8, device="cuda:0")
        gs.reads(spec, P, 1000, n, a, b if paired else None)
        torch.cuda.synchronize()
        exp = cs.reads(spec, P, 1000, n)
        if paired:
            assert np.array_equal(a.cpu().numpy(), exp[0]) and np.array_equal(b.cpu().numpy(), exp[1])
        else:
            assert np.array_equal(a.cpu().numpy(), exp)


@pytest.mark.parametrize("shards", [1, 3])
def test_builder_device_sources_key_shards_streaming_table(tmp_path, shards):
    """Targets generated in HBM -> mc_build_add_target_device (lane sketcher, per-shard pair selection) -> mc_build_table_* must give
    the table the host-source builder gives (its files are read by the oracle), and both the table the oracle builds itself."""
    spec = synthdb.phylogeny(4, 2, 3, 30_000, 90_000, seed=21 + shards)
    # window-count edge cases of the chunk records (256 windows each; a target's last record: up to 256 full windows + the tail)
    edge = [255 * 112 + 127, 255 * 112 + 127 + 15, 255 * 112 + 127 + 16, 255 * 112 + 127 + 67, 511 * 112 + 127 + 30, 256 * 112 + 127,
            126, 127, 128, 16, 15, 254 * 112 + 127 + 111]
    spec.targets["length"][:len(edge)] = edge
    cs = synthdb.CpuSynth()
    K = 3
    db, info = synthdb.build_database(spec, shards=shards, chunk_bytes=400_000, max_candidates=K)     # several flush groups
    # the same collection through the host path, written as database files
    bld = api.Builder(target_id_bytes=4, max_candidates=K)
    for t in range(len(spec.targets)):
        bld.add_target(cs.target(spec, t), f"SYN_{t:06d}.1", parent_taxid=spec.parent_taxid(t))
    bld.finish(load=False)
    name = str(tmp_path / "syn")
    bld.write(name, spec.taxa())
    nk, nv = bld.counts()
    bld.free()
    assert db.n_locations == nv
    ofile = cpuref.oracle().open(name)
    oself = scale_util.oracle_database(spec, None, threads=4)
    k1, s1, _, _ = ofile.part_arrays(); k2, s2, _, _ = oself.part_arrays()
    assert len(k1) == len(k2) == nk and int(s1.sum()) == int(s2.sum()) == nv
    for f in k1[::53]:
        assert np.array_equal(ofile.lookup(int(f)), oself.lookup(int(f)))
    P = synthdb.read_params(spec, 5)
    reads = cs.reads(spec, P, 0, 1500)
    rl = [bytes(r[:150]) for r in reads] + [bytes(synth.random_genome(np.random.default_rng(1), 150)) for _ in range(50)]
    cands, counts, _ = db.query(rl)
    for i, r in enumerate(rl):
        _, e = oself.query(r, b"", K, 0, 0)
        _check(cands[i], e, K, i)
        _, e2 = ofile.query(r, b"", K, 0, 0)
        assert [tuple(x)[1:] for x in e.tolist()] == [tuple(x)[1:] for x in e2.tolist()]
    db.close(); ofile.close(); oself.close()


@pytest.mark.parametrize("lowest,K,store", [(0, 1, 4), (0, 2, 8), (0, 4, 4), (4, 2, 4), (6, 3, 8)])
def test_big_cands_filtered_lists_against_oracle(monkeypatch, lowest, K, store):
    """k = 10: the feature space is so small that every bucket fills up with locations of unrelated targets, as 32-bit features do at
    RefSeq scale: a 150 bp read collects 1000 .. 5000 locations, a handful of them on its true targets -- the lists big_cands_kernel
    filters by target before counting.  Reads of the collection (strong candidates), random reads and reads of a single window
    (fewer than K targets with two hits: the open places go to the smallest single-hit targets), sequence level and merged."""
    monkeypatch.setenv("MC_BIG_MIN", "0")
    monkeypatch.setenv("MC_COMPACT_LOCATIONS", "1" if store == 4 else "0")      # location store: 4 / 8 bytes per location
    spec = synthdb.phylogeny(120, 2, 3, 40_000, 60_000, seed=100 + lowest + K)
    sk = dict(kmerlen=10, sketchlen=16, winlen=121, winstride=112)
    db, info = synthdb.build_database(spec, shards=2, max_candidates=K, **sk)
    assert db.table_layout()["location_bytes"] == store
    db.set_lineages(spec.lineages())
    odb = scale_util.oracle_database(spec, None, threads=THREADS, with_lineages=True, k=10, s=16, w=121, stride=112)
    cs = synthdb.CpuSynth()
    P = synthdb.read_params(spec, 77, sub_rate=0.02)
    reads = [bytes(r[:150]) for r in cs.reads(spec, P, 0, 2500)]
    rng = np.random.default_rng(5)
    reads += [bytes(synth.random_genome(rng, 150)) for _ in range(700)]
    reads += [r[:70] for r in reads[:300]]
    cands, counts, _ = db.query(reads, lowest=lowest)
    assert np.mean(counts > 1024) > 0.5, (np.mean(counts > 1024), counts.max())
    for i, r in enumerate(reads):
        _, e = odb.query(r, b"", K, lowest, 0)
        _check(cands[i], e, K, (i, counts[i]))
    # every shipped variant of the lane path's kernels (sketch + probe in one kernel or apart, both bucket
    # fetch schemes, filter and counting apart): the same candidates
    for v in scale_util.each_variant(db):
        cv, cnt_v, _ = db.query(reads, lowest=lowest)
        assert np.array_equal(cnt_v, counts), v
        for f in ("tgt", "hits", "beg", "end"):
            bad = np.nonzero((cv[f] != cands[f]).any(axis=1))[0]
            assert len(bad) == 0, (v, f, bad[:5], cv[bad[:2]], cands[bad[:2]])
    # Mode K's two halves on the same lists: the shard side hands the lists over as they are (MC_WANT_PARTIAL_HITS, lane path), the
    # owner side unites them (one source here) and sends the long ones through the same filter with the union buffer as its table
    import torch
    sub = [r for r in reads[:1500] if len(r) == 150]
    buf = np.zeros(len(sub) * 152 + 16, dtype=np.uint8)
    for i, r in enumerate(sub):
        buf[i * 152:i * 152 + 150] = np.frombuffer(r, dtype=np.uint8)
    dseq = torch.from_numpy(buf).cuda()
    qinfo = torch.zeros((len(sub), 4), dtype=torch.int32, device="cuda")
    qinfo[:, 0] = torch.arange(len(sub), dtype=torch.int32, device="cuda") * 152; qinfo[:, 1] = 150; qinfo[:, 2] = qinfo[:, 0]
    res = db.query_device(dseq.data_ptr(), qinfo.data_ptr(), len(sub), len(sub) * 152, max_win_uniform=3, want_partial_hits=True)
    off = torch.zeros(len(sub) + 1, dtype=torch.int64, device="cuda")
    db.copy_results(off.data_ptr(), res.hit_offsets, (len(sub) + 1) * 8); db.synchronize()
    hits = torch.zeros(int(off[-1]), dtype=torch.int64, device="cuda")
    db.copy_results(hits.data_ptr(), res.hits, hits.numel() * 8); db.synchronize()
    cnt = (off[1:] - off[:-1]).to(torch.int32).contiguous()
    r2 = db.candidates_from_partial_hits(cnt.data_ptr(), hits.data_ptr(), hits.numel(), len(sub), 1, max_win_uniform=3, lowest=lowest)
    oc = torch.zeros((len(sub), K, 4), dtype=torch.int32, device="cuda")
    db.copy_results(oc.data_ptr(), r2.cands, len(sub) * K * 16); db.synchronize()
    oc = oc.cpu().numpy().view(np.uint32)
    idx = {r: i for i, r in enumerate(reads)}
    assert int((cnt > 1024).sum()) > len(sub) // 3
    for j, r in enumerate(sub):
        g = cands[idx[r]]
        assert [tuple(int(x) for x in oc[j, k]) if g[k]["hits"] else 0 for k in range(K)] == \
               [(int(g[k]["tgt"]), int(g[k]["hits"]), int(g[k]["beg"]), int(g[k]["end"])) if g[k]["hits"] else 0 for k in range(K)], (j, oc[j], g)
    # pairs: twice the features (up to 64 entries), maxWindowsInRange 4 / 5
    mates = [bytes(synth.revcomp(np.frombuffer(r, dtype=np.uint8)))[:110] for r in reads[:800]]
    pc, pcounts, _ = db.query(reads[:800], mates, lowest=lowest, insert_max=400)
    for i in range(800):
        _, e = odb.query(reads[i], mates[i], K, lowest, 400)
        _check(pc[i], e, K, ("pair", i, pcounts[i]))
    db.close(); odb.close()


@pytest.mark.parametrize("lowest,K,store", [(0, 2, 4), (0, 4, 8), (4, 3, 4)])
def test_big_cands_strain_rich_lists_overflow_paths(monkeypatch, lowest, K, store):
    """4 species x 110 strains: nearly every location of a read's list lies on a target with many hits, so the filter keeps almost
    everything: lists beyond the first instance's 512 go on to the second (1024) and from there to the wave kernel; ties between
    near-identical strains everywhere."""
    monkeypatch.setenv("MC_BIG_MIN", "0")
    monkeypatch.setenv("MC_COMPACT_LOCATIONS", "1" if store == 4 else "0")
    spec = synthdb.phylogeny(2, 2, 110, 20_000, 24_000, seed=300 + lowest + K, div_strain=(0.002, 0.01))
    db, info = synthdb.build_database(spec, shards=1, max_candidates=K)
    assert db.table_layout()["location_bytes"] == store
    db.set_lineages(spec.lineages())
    odb = scale_util.oracle_database(spec, None, threads=THREADS, with_lineages=True)
    cs = synthdb.CpuSynth()
    P = synthdb.read_params(spec, 78)
    reads = [bytes(r[:150]) for r in cs.reads(spec, P, 0, 2000)]
    for j, L in enumerate((20, 24, 30, 40, 60, 100)):                               # fewer k-mers, fewer features: shorter lists
        reads += [bytes(r[:L]) for r in cs.reads(spec, P, 5000 + 300 * j, 300)]
    cands, counts, _ = db.query(reads, lowest=lowest)
    assert np.mean(counts > 1024) > 0.3 and np.any((counts > 256) & (counts <= 512)) and np.any((counts > 512) & (counts <= 1024)), np.percentile(counts, [5, 50, 95])
    for i, r in enumerate(reads):
        _, e = odb.query(r, b"", K, lowest, 0)
        _check(cands[i], e, K, (i, counts[i]))
    db.close(); odb.close()


def test_two_gbp_database_auto_quad_against_oracle():
    """BASELINE configs[2] in small: a 2.1 Gbp phylogeny (100 genera x 2 species x 3 strains x 3.5 Mbp, uint32 targets), built in 2 key
    shards through the streaming table; the bucket array (> 1 GiB) makes probe_cands choose the quad-cooperative fetch BY ITSELF.
    20 000 single reads and 8 000 pairs against the oracle, which builds the buckets of the sample's features on the host cores."""
    assert "MC_QUAD_LOOKUP" not in os.environ and "MC_BIG_MIN" not in os.environ
    spec = synthdb.phylogeny(100, 2, 3, 3_400_000, 3_600_000, seed=2100)
    K = 2
    db, info = synthdb.build_database(spec, shards=2, max_candidates=K, max_load_factor=0.3)
    assert info["bases"] > 2_000_000_000
    import torch
    gen = synthdb.GpuSynth(0)
    n1, n2 = 20_000, 8_000
    P1 = synthdb.read_params(spec, 3100)
    P2 = synthdb.read_params(spec, 4100, paired=True)
    a = torch.zeros((n1, P1.row_bytes), dtype=torch.uint8, device="cuda:0")
    m1 = torch.zeros((n2, P2.row_bytes), dtype=torch.uint8, device="cuda:0")
    m2 = torch.zeros((n2, P2.row_bytes), dtype=torch.uint8, device="cuda:0")
    gen.reads(spec, P1, 0, n1, a)
    gen.reads(spec, P2, 0, n2, m1, m2)
    torch.cuda.synchronize()
    singles = [bytes(r[:150]) for r in a.cpu().numpy()]
    p1 = [bytes(r[:150]) for r in m1.cpu().numpy()]
    p2 = [bytes(r[:150]) for r in m2.cpu().numpy()]
    # features of the sample (oracle sketcher) -> the oracle's own restricted build
    odb = scale_util.oracle_database(spec, scale_util.sample_features(singles + p1 + p2), threads=THREADS)
    st = db.info()
    assert st[5] == len(spec.targets)
    assert db.table_layout()["location_bytes"] == 4                 # 600 targets x 32 000 windows: the compact store, chosen by the builder
    cands, counts, _ = db.query(singles)
    pc, pcounts, _ = db.query(p1, p2, insert_max=0)
    for i in range(n1):
        _, e = odb.query(singles[i], b"", K, 0, 0)
        _check(cands[i], e, K, (i, counts[i]))
    for i in range(n2):
        _, e = odb.query(p1[i], p2[i], K, 0, 0)
        _check(pc[i], e, K, ("pair", i, pcounts[i]))
    assert np.mean(cands[:, 0]["hits"] >= 8) > 0.9
    # two batches in flight from two host threads: the context's second pipe (MC_SECOND_PIPE) next to the first, each on its own stream
    import threading
    rows = a[:, :152].contiguous() if P1.row_bytes >= 152 else None
    assert rows is not None
    seqs = [torch.cat([rows[:n1 // 2].reshape(-1), torch.zeros(16, dtype=torch.uint8, device="cuda:0")]),
            torch.cat([rows[n1 // 2:].reshape(-1), torch.zeros(16, dtype=torch.uint8, device="cuda:0")])]
    m = n1 // 2
    qinfo = torch.zeros((m, 4), dtype=torch.int32, device="cuda:0")
    qinfo[:, 0] = torch.arange(m, dtype=torch.int32, device="cuda:0") * 152; qinfo[:, 1] = 150; qinfo[:, 2] = qinfo[:, 0]
    outs = [torch.zeros((m, K, 4), dtype=torch.int32, device="cuda:0") for _ in range(2)]
    streams = [torch.cuda.Stream(device="cuda:0") for _ in range(2)]
    torch.cuda.synchronize()
    errs = []

    def worker(t):
        try:
            for _ in range(3):
                r = db.query_device(seqs[t].data_ptr(), qinfo.data_ptr(), m, m * 152, max_win_uniform=3, stream=streams[t].cuda_stream, second_pipe=(t == 1))
                db.copy_results(outs[t].data_ptr(), r.cands, m * K * 16, stream=streams[t].cuda_stream)
                streams[t].synchronize()
        except Exception as e:                                      # noqa: BLE001
            errs.append(e)
    th = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errs, errs
    got = torch.cat(outs).cpu().numpy().view(np.uint32)
    for i in range(n1):
        for k in range(K):
            g = cands[i][k]
            exp = (int(g["tgt"]), int(g["hits"]), int(g["beg"]), int(g["end"])) if g["hits"] else None
            assert (tuple(int(x) for x in got[i, k]) if got[i, k, 1] else None) == exp, (i, k, got[i], g)
    db.close(); odb.close()


def test_compact_store_holds_shapes_that_do_not_fit_32_bits_as_target_and_window():
    """32 800 targets (16 bits) of which a few have 75 000 windows (17 bits): round 2's (target << bits) | window form could not hold
    this collection in 4 bytes; the global window numbers do.  Reads over all targets against the oracle's restricted build; the large
    genomes are strain groups, so their reads take the filtered path's kernels (MC_BIG_MIN=0: every list above 64)."""
    import torch
    spec = synthdb.phylogeny(8200, 2, 2, 1_000, 3_000, seed=77, big_fraction=0.0012, big_len=(8_000_000, 8_400_000))
    lens = spec.targets["length"].astype(np.int64)
    assert len(lens) == 32_800 and (lens // 112).max() > 65_536 and (lens > 1_000_000).sum() >= 8
    K = 2
    db, _ = synthdb.build_database(spec, shards=1, max_candidates=K)
    lay = db.table_layout()
    assert lay["location_bytes"] == 4, lay
    db.set_tuning("big_min", 0)
    gen = synthdb.GpuSynth(0)
    n = 30_000
    P = synthdb.read_params(spec, 77)
    a = torch.zeros((n, P.row_bytes), dtype=torch.uint8, device="cuda:0")
    gen.reads(spec, P, 0, n, a)
    torch.cuda.synchronize()
    reads = [bytes(r[:150]) for r in a.cpu().numpy()]
    # reads are drawn uniformly over the TARGETS: add reads of the large genomes' windows beyond 65 536 on purpose
    cs = synthdb.CpuSynth()
    big = np.flatnonzero(lens > 1_000_000)
    rng = np.random.default_rng(5)
    for t in big[:12]:
        for _ in range(40):
            st = int(rng.integers(66_000 * 112, int(lens[t]) - 200))
            reads.append(bytes(cs.target(spec, int(t), st, 150)))
    odb = scale_util.oracle_database(spec, scale_util.sample_features(reads), threads=THREADS)
    cands, counts, _ = db.query(reads)
    far = 0
    for i, r in enumerate(reads):
        _, e = odb.query(r, b"", K, 0, 0)
        _check(cands[i], e, K, (i, counts[i]))
        far += int(len(e) > 0 and e[0]["end"] > 65_536)
    assert far > 300, far                                          # candidates whose window numbers need 17 bits
    db.close(); odb.close()


@pytest.fixture(scope="module")
def table33():
    """8 800 targets / 33 Gbp of the bench collection's shape (440 genera x 4 species x 5 strains, 2.5 - 5 Mbp), built in 4 key shards:
    more than 2^32 locations.  One build for the tests below."""
    assert "MC_BIG_MIN" not in os.environ and "MC_COMPACT_LOCATIONS" not in os.environ
    spec = synthdb.phylogeny(440, 4, 5, 2_500_000, 5_000_000, seed=3100)
    db, info = synthdb.build_database(spec, shards=4, max_candidates=2)
    yield spec, db
    db.close()


def test_long_reads_length_distribution_against_oracle(table33):
    """BASELINE configs[4]'s reads on the 33 Gbp table: 2 000 single reads, lengths log-normal around a median of 480 bp, clipped to
    200 .. 19 000 (README.md:5), 7.5 % substitutions, seed 5100; maxWindowsInRange = 2 + length / 112 (candidate_structs.hpp:143-145): up to
    171.  They collect hundreds to tens of thousands of locations; beyond 512 bp they are sketched and probed by the chunk lanes, filtered by
    gw_filter_stream_kernel, sorted (gw_sort.hip) and scanned (gw_sorted_cands_kernel).  Every candidate against the oracle's restricted
    build."""
    import torch
    spec, db = table33
    K = 2
    n, Lmax = 2000, 19_000
    rng = np.random.default_rng(5100)
    lens = np.clip(np.exp(rng.normal(np.log(480.0), 0.95, n)), 200, Lmax).astype(np.int64)
    lens[:4] = (Lmax, 12_345, 200, 513)                       # the ends of the range are in whatever the draw says
    P = synthdb.read_params(spec, 5100, read_len=Lmax, sub_rate=0.075)
    rows = torch.zeros((n, P.row_bytes), dtype=torch.uint8, device="cuda:0")
    synthdb.GpuSynth(0).reads(spec, P, 0, n, rows)
    torch.cuda.synchronize()
    host = rows.cpu().numpy()
    reads = [bytes(host[i, :int(lens[i])]) for i in range(n)]
    assert np.median(lens) < 600 and lens.max() == Lmax and (lens > 2000).sum() > 50
    odb = scale_util.oracle_database(spec, scale_util.sample_features(reads), threads=THREADS)
    db.timing(True); db.timing_reset()
    cands0, counts0, _ = db.query(reads)
    assert db.timing_get("gw_sort")[1] > 0 and db.timing_get("gw_sorted_cands")[1] > 0      # the sorted path ran
    assert counts0.max() > 20_000, counts0.max()
    db.timing(False)
    cands, counts = cands0, counts0
    db.set_tuning("gw_big_h", 4096)                               # the fine-block instance of the stream filter from 4 096 locations on (default: 32 768)
    cands_fine, _, _ = db.query(reads)
    db.set_tuning("gw_big_h", 32768)
    for i in range(n):
        _, e = odb.query(reads[i], b"", K, 0, 0)
        _check(cands[i], e, K, (i, int(lens[i]), counts[i]))
        _check(cands0[i], e, K, ("sorted", i, int(lens[i]), counts[i]))
        _check(cands_fine[i], e, K, ("fine", i, int(lens[i]), counts[i]))
    odb.close()


def test_more_than_2_32_locations_against_oracle(table33):
    """A table beyond 2^32 locations (location offsets, list store and the builder's streaming shards all past 32 bits); 20 000 reads and
    4 000 pairs against the oracle's restricted build.  Lists of 300 - 500 locations per read: the filtered path at its short end."""
    spec, db = table33
    K = 2
    st = db.info()
    assert st[5] == 8800 and st[7] > (1 << 32), st
    lay = db.table_layout()
    assert lay["location_bytes"] == 4 and lay["list_locations"] > (1 << 32), lay
    import torch
    gen = synthdb.GpuSynth(0)
    n1, n2 = 20_000, 4_000
    P1 = synthdb.read_params(spec, 3100)
    P2 = synthdb.read_params(spec, 4100, paired=True)
    a = torch.zeros((n1, P1.row_bytes), dtype=torch.uint8, device="cuda:0")
    m1 = torch.zeros((n2, P2.row_bytes), dtype=torch.uint8, device="cuda:0")
    m2 = torch.zeros((n2, P2.row_bytes), dtype=torch.uint8, device="cuda:0")
    gen.reads(spec, P1, 0, n1, a)
    gen.reads(spec, P2, 0, n2, m1, m2)
    torch.cuda.synchronize()
    singles = [bytes(r[:150]) for r in a.cpu().numpy()]
    p1 = [bytes(r[:150]) for r in m1.cpu().numpy()]
    p2 = [bytes(r[:150]) for r in m2.cpu().numpy()]
    odb = scale_util.oracle_database(spec, scale_util.sample_features(singles + p1 + p2), threads=THREADS)
    cands, counts, _ = db.query(singles)
    pc, pcounts, _ = db.query(p1, p2, insert_max=0)
    assert np.mean(counts > 256) > 0.5, np.percentile(counts, [5, 50, 95])
    for i in range(n1):
        _, e = odb.query(singles[i], b"", K, 0, 0)
        _check(cands[i], e, K, (i, counts[i]))
    for i in range(n2):
        _, e = odb.query(p1[i], p2[i], K, 0, 0)
        _check(pc[i], e, K, ("pair", i, pcounts[i]))
    odb.close()


@pytest.mark.parametrize("lowest,K,store", [(0, 2, 4), (0, 3, 8), (4, 2, 4)])
def test_filter_second_instance_reads_of_five_to_ten_windows(monkeypatch, lowest, K, store):
    """Reads of 300 .. 512 bp and 2 x 250 bp pairs find 65 .. 160 features: more entries than big_filter_kernel's one per lane -- its
    second instance (three per lane, same pool slices after the first).  12 strains per species: lists of 200 .. 2000 locations, filtered
    lists on both sides of the counting kernels' limits; mixed with 150 bp reads (first instance) in the same batches."""
    monkeypatch.setenv("MC_BIG_MIN", "0")
    monkeypatch.setenv("MC_COMPACT_LOCATIONS", "1" if store == 4 else "0")
    spec = synthdb.phylogeny(4, 2, 12, 30_000, 40_000, seed=500 + lowest + K, div_strain=(0.002, 0.01))
    db, info = synthdb.build_database(spec, shards=1, max_candidates=K)
    assert db.table_layout()["location_bytes"] == store
    db.set_lineages(spec.lineages())
    odb = scale_util.oracle_database(spec, None, threads=THREADS, with_lineages=True)
    cs = synthdb.CpuSynth()
    reads = []
    for j, L in enumerate((300, 380, 450, 500, 512, 150)):
        P = synthdb.read_params(spec, 900 + j, read_len=L)
        reads += [bytes(r[:L]) for r in cs.reads(spec, P, 0, 500)]
    rng = np.random.default_rng(9)
    reads += [bytes(synth.random_genome(rng, 480)) for _ in range(100)]
    order = rng.permutation(len(reads))
    reads = [reads[i] for i in order]
    db.timing(True); db.timing_reset()
    cands, counts, _ = db.query(reads, lowest=lowest)
    if store == 4:
        assert db.timing_get("gw_filter_stream")[1] > 0 and (db.timing_get("gw_filter_count")[1] > 0 or db.timing_get("gw_filter")[1] > 0)
    else:
        assert db.timing_get("big_filter_2")[1] > 0 and db.timing_get("big_filter")[1] > 0
    assert np.mean(counts > 256) > 0.5, np.percentile(counts, [5, 50, 95])
    for i, r in enumerate(reads):
        _, e = odb.query(r, b"", K, lowest, 0)
        _check(cands[i], e, K, (i, len(r), counts[i]))
    P2 = synthdb.read_params(spec, 950, read_len=250, paired=True)
    m1, m2 = cs.reads(spec, P2, 0, 1200)
    p1 = [bytes(r[:250]) for r in m1]; p2 = [bytes(r[:250]) for r in m2]
    pc, pcounts, _ = db.query(p1, p2, lowest=lowest, insert_max=0)
    for i in range(len(p1)):
        _, e = odb.query(p1[i], p2[i], K, lowest, 0)
        _check(pc[i], e, K, ("pair", i, pcounts[i]))
    db.timing(False)
    db.close(); odb.close()


@pytest.mark.parametrize("K,lowest,seed", [(2, 0, 11), (4, 0, 12), (1, 4, 13), (3, 4, 14), (2, 6, 15)])
def test_sorted_path_strain_rich_long_reads_against_oracle(tmp_path, K, lowest, seed):
    """The sorted path (stream filter -> segmented sort -> gw_sorted_cands_kernel) where its scan has the most to get wrong: long reads
    (700 .. 19 000 bp, window ranges of 8 .. 171) on groups of up to 60 close strains -- filtered lists of tens of thousands of numbers
    with dozens of targets, target runs that cross the 64-number chunks and the sixteen runs of the block pass, ties between strains,
    taxon merging at species and genus level, K = 1 .. 4 -- every candidate against the oracle."""
    from metacache_amd import synth
    rng = np.random.default_rng(seed)
    genomes, parents = [], []
    for sp, (nst, size, div) in enumerate([(60, 24_000, 0.004), (25, 30_000, 0.01), (9, 21_000, 0.03), (1, 40_000, 0.0), (1, 26_000, 0.0)]):
        base = synth.random_genome(rng, size)
        for st in range(nst):
            genomes.append(synth.mutate(rng, base, div) if st else base)
            parents.append(1000 + sp)
    bld = api.Builder(target_id_bytes=4, max_candidates=K)
    for i, g in enumerate(genomes):
        bld.add_target(g, f"S{i:04d}.1", parent_taxid=parents[i])
    name = str(tmp_path / "strains")
    bld.finish(load=False)
    bld.write(name, [(1, 1, 20, "root"), (500, 1, 6, "genus a"), (501, 1, 6, "genus b")] + [(1000 + i, 500 + i % 2, 4, f"sp{i}") for i in range(5)])
    bld.free()
    reads = []
    for i in range(240):
        g = genomes[int(rng.integers(len(genomes)))]
        # (the last 80: reads of 7 .. 20 windows on the small strain groups -- lists gw_count_block_kernel's tables hold)
        L = int(min(g.size - 1, rng.choice([700, 1500, 3000, 6000, 12_000, 19_000]) if i < 160 else rng.choice([700, 900, 1200, 1500, 2200])))
        if i >= 160: g = genomes[int(rng.integers(85, len(genomes)))]
        p = int(rng.integers(0, g.size - L))
        r = synth.mutate(rng, g[p:p + L], float(rng.choice([0.0, 0.02, 0.075])))
        if i % 5 == 0:                                            # chimeras: two genomes' pieces in one read
            g2 = genomes[int(rng.integers(len(genomes)))]
            r = np.concatenate([r[: L // 2], g2[: L // 2]])
        reads.append(bytes(synth.revcomp(r) if rng.random() < 0.5 else r))
    odb = cpuref.oracle().open(name)
    db = api.Database.open(name, max_candidates=K, slot_max_queries=1 << 10, slot_max_chars=1 << 23)
    assert db.table_layout()["location_bytes"] == 4
    db.timing(True); db.timing_reset()
    cands0, counts, _ = db.query(reads, lowest=lowest)
    assert db.timing_get("gw_sorted_cands")[1] > 0 and counts.max() > 30_000, counts.max()
    db.timing(False)
    cands = cands0
    # the stream filter's fine-block instance (blocks of 2^A >= D numbers, the neighbour blocks asked as well; by default for reads beyond
    # 32 768 locations) for EVERY read of the stream filter, and none at all: the same candidates
    db.set_tuning("gw_big_h", 2048)
    cands_fine, _, _ = db.query(reads, lowest=lowest)
    db.set_tuning("gw_big_h", 0)
    cands_coarse, _, _ = db.query(reads, lowest=lowest)
    db.close()
    # batches whose filter grid is NOT a multiple of four blocks (the fine-block instance of sixteen waves owns the pool slices of four
    # such blocks; its last block fewer) and batches of fewer than thirteen reads (fewer slices than one such block has waves): batch
    # sizes 241 (61 blocks), 9 (3 blocks), 1 -- with the fine-block instance on every read of the stream filter
    if K == 2:
        for bs in (241, 9, 1):
            dbs = api.Database.open(name, max_candidates=K, slot_max_queries=bs, slot_max_chars=1 << 23)
            dbs.set_tuning("gw_big_h", 2048)
            sub = reads if bs > 1 else reads[:40]
            cs, _, _ = dbs.query(sub, lowest=lowest)
            dbs.close()
            for i, r in enumerate(sub):
                _, e = odb.query(r, b"", K, lowest, 0)
                _check(cs[i], e, K, ("batch size", bs, i, len(r)))
    for i, r in enumerate(reads):
        _, e = odb.query(r, b"", K, lowest, 0)
        _check(cands[i], e, K, (i, len(r), counts[i]))
        _check(cands0[i], e, K, ("sorted", i, len(r), counts[i]))
        _check(cands_fine[i], e, K, ("fine", i, len(r), counts[i]))
        _check(cands_coarse[i], e, K, ("coarse", i, len(r), counts[i]))
    odb.close()
