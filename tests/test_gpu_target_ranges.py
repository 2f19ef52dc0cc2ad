"""GPU: ONE database file cut into contiguous target ranges at load ("Mode T": mc_config.target_shard_index / _count) -- every range is
a context of its own that answers with the unchanged single-table path; the per-range top lists merged in range order
(mc_partset_open with target_shard_count > 1 -> mc_merge_part_candidates) must be the whole table's lists, bit for bit: the
candidates of a read never span two targets (candidate_generation.hpp:96-150 starts a new candidate with every new target), and
the ranges follow the target order the whole table's insertion order has (candidate_structs.hpp:170-213)."""
import ctypes as C

import numpy as np
import pytest

import cpuref
from metacache_amd import api, synthdb

pytestmark = pytest.mark.gpu


def _same(got, exp, tag):
    used = exp["hits"] > 0
    for f in ("hits", "beg", "end"):
        assert np.array_equal(got[f], exp[f]), (tag, f)
    assert np.array_equal(got["tgt"][used], exp["tgt"][used]), tag


@pytest.mark.parametrize("db", ["toy16", "toy32"])
@pytest.mark.parametrize("ranges,resident,K,lowest", [(2, 2, 2, 0), (3, 2, 3, 4), (5, 5, 1, 0), (7, 3, 4, 6), (64, 64, 2, 0)])
def test_ranges_of_the_goldens_equal_the_whole_table(golden, db, ranges, resident, K, lowest):
    single, p1, p2 = golden.reads()
    name = golden.db_path(db)
    whole = api.Database.open(name, max_candidates=K)
    ntargets = whole.n_targets
    cw, _, _ = whole.query(single, lowest=lowest)
    cp, _, _ = whole.query(p1, p2, lowest=lowest, insert_max=700)
    nloc = whole.n_locations
    whole.close()
    # the ranges themselves: contiguous, in order, together all targets and all locations (more ranges than targets: empty ones)
    lo_expected, total = 0, 0
    for r in range(ranges):
        part = api.Database.open(name, max_candidates=K, target_shard_index=r, target_shard_count=ranges)
        lo, hi = part.target_range()
        if hi > lo:
            assert lo == lo_expected, (r, lo, hi)
            lo_expected = hi
        total += part.n_locations
        part.close()
    assert lo_expected == ntargets and total == nloc
    ps = api.PartSet(name, resident=resident, max_candidates=K, target_shard_count=ranges, slot_max_queries=700, slot_max_chars=1 << 17)
    info = ps.info()
    assert info["parts"] == ranges and info["resident"] == resident
    _same(ps.classify(single, lowest=lowest), cw, "single")
    _same(ps.classify(p1, p2, lowest=lowest, insert_max=700), cp, "pairs")
    ps.close()


@pytest.mark.parametrize("rules", [dict(max_locations_per_feature=3), dict(remove_overpopulated=6), dict(max_locations_per_feature=5, remove_overpopulated=40)])
def test_load_rules_look_at_the_whole_bucket(golden, rules):
    """-max-locations-per-feature / -remove-overpopulated-features (host_hashmap.hpp:454-495) cut the FILE's bucket, as on the whole table,
    before the range is taken out of it"""
    single, _, _ = golden.reads()
    name = golden.db_path("toy32")
    K = 2
    whole = api.Database.open(name, max_candidates=K, **rules)
    cw, _, _ = whole.query(single)
    nloc = whole.n_locations
    whole.close()
    assert nloc < api.Database.open(name, max_candidates=K).n_locations     # (the rule does cut something)
    total = 0
    for r in range(3):
        part = api.Database.open(name, max_candidates=K, target_shard_index=r, target_shard_count=3, **rules)
        total += part.n_locations
        part.close()
    assert total == nloc
    ps = api.PartSet(name, resident=3, max_candidates=K, target_shard_count=3, **rules)
    _same(ps.classify(single), cw, rules)
    ps.close()


@pytest.mark.parametrize("pipelined", ["1", "0"])
def test_ranges_of_a_filtered_list_table(tmp_path, monkeypatch, pipelined):
    """k = 10: every read collects thousands of locations of unrelated targets (the lists the filter kernels cut by target before they
    count, as 32-bit features give at RefSeq scale) -- 4 ranges of the written file against the whole file and the oracle; both loaders
    (the pipelined one and the batch-by-batch one)"""
    monkeypatch.setenv("MC_BIG_MIN", "0")
    monkeypatch.setenv("MC_LOAD_PIPELINE", pipelined)
    spec = synthdb.phylogeny(60, 2, 3, 30_000, 70_000, seed=314)
    sk = dict(kmerlen=10, sketchlen=16, winlen=121, winstride=112)
    name = str(tmp_path / "syn")
    K = 3
    db, _ = synthdb.build_database(spec, shards=1, max_candidates=K, write_to=name, **sk)
    db.close()
    cs = synthdb.CpuSynth()
    P = synthdb.read_params(spec, 9, sub_rate=0.02)
    reads = [bytes(r[:150]) for r in cs.reads(spec, P, 0, 1200)]
    whole = api.Database.open(name, max_candidates=K)
    layout = whole.table_layout()
    for lowest in (0, 4):
        cw, _, _ = whole.query(reads, lowest=lowest)
        ps = api.PartSet(name, resident=4, max_candidates=K, target_shard_count=4)
        _same(ps.classify(reads, lowest=lowest), cw, lowest)
        ps.close()
    odb = cpuref.oracle().open(name)
    cw, _, _ = whole.query(reads)
    for i in range(0, len(reads), 7):
        _, e = odb.query(reads[i], b"", K, 0, 0)
        e = e[:K]
        assert [(int(x["tgt"]), int(x["hits"]), int(x["beg"]), int(x["end"])) for x in e] == \
               [(int(c["tgt"]), int(c["hits"]), int(c["beg"]), int(c["end"])) for c in cw[i][:len(e)]], i
    odb.close()
    # a range's store is its share of the locations, not the file's
    part = api.Database.open(name, max_candidates=K, target_shard_index=1, target_shard_count=4)
    assert part.n_locations < 0.3 * whole.n_locations and part.table_layout()["list_locations"] < 0.7 * layout["list_locations"]
    c1, _, _ = part.query(reads)
    lay1 = part.table_layout()
    part.close()
    if pipelined == "1":
        # an estimate of the range's store that is too small: the file is counted to its end and loaded again with the exact size
        monkeypatch.setenv("MC_TARGET_STORE_MARGIN", "0.5")
        part = api.Database.open(name, max_candidates=K, target_shard_index=1, target_shard_count=4)
        c2, _, _ = part.query(reads)
        lay2 = part.table_layout()                                # (with the exact numbers: lines of their own would take 4 x the plain store here -> plain)
        assert np.array_equal(c1, c2) and lay2["list_align"] == 1 and lay2["list_locations"] < lay1["list_locations"]
        part.close()
    whole.close()


def test_refusals(golden):
    name = golden.db_path("toy16")
    with pytest.raises(api.McError, match="target_shard_index"):
        api.Database.open(name, target_shard_index=4, target_shard_count=4)
    with pytest.raises(api.McError, match="cannot be combined"):
        api.Database.open(name, target_shard_index=0, target_shard_count=2, key_shard_index=0, key_shard_count=2)
    with pytest.raises(api.McError, match="single part"):
        api.Database.open(golden.db_path("toy32p2"), target_shard_index=0, target_shard_count=2)
    # host arrays (mc_load_batch) are not cut: the ranges come from the target metadata of a database file
    L = api.lib()
    cfg = api.default_config(kmerlen=16, sketchlen=16, winlen=127, winstride=112, max_candidates=2, target_id_bytes=4, target_shard_count=2)
    h = C.c_void_p()
    assert L.mc_create(C.byref(cfg), C.byref(h)) == 0
    assert L.mc_load_begin(h, 0, 10, 10) != 0
    L.mc_destroy(h)
    # one part of a partitioned database can be cut as well
    ps = api.PartSet(golden.db_path("toy32p2"), resident=2, max_candidates=2, single_part=1, target_shard_count=2)
    assert ps.info()["parts"] == 2
    ps.close()


def test_target_ranges_at_midscale_filtered_regime():
    """A 15 Gbp cut of the bench collection (195 locations per 150 bp read: the filtered path's regime) written as ONE database file, then
    opened as 4 target ranges: singles, pairs and long reads (window ranges up to 171, sorted lists), sequence level and species level,
    against the single table holding everything -- whose results tests/test_gpu_reference_midscale.py holds against the reference itself
    on the same collection.  The ranges' tables are sized by what the loader's sample of the file says they hold."""
    import os
    import torch
    shm = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
    name = os.path.join(shm, f"mc_ranges_{os.getpid()}")
    spec = synthdb.phylogeny(200, 4, 5, 2_500_000, 5_000_000, seed=3100)
    K = 2
    n1, n2, n3 = 40_000, 8_000, 600
    try:
        db, _ = synthdb.build_database(spec, shards=2, max_candidates=K, write_to=name)
        assert db.table_layout()["location_bytes"] == 4
        whole_layout = db.table_layout()
        nloc = db.n_locations
        gen = synthdb.GpuSynth(0)
        P1 = synthdb.read_params(spec, 3100)
        P2 = synthdb.read_params(spec, 4100, paired=True)
        a = torch.zeros((n1, P1.row_bytes), dtype=torch.uint8, device="cuda:0")
        m1 = torch.zeros((n2, P2.row_bytes), dtype=torch.uint8, device="cuda:0")
        m2 = torch.zeros((n2, P2.row_bytes), dtype=torch.uint8, device="cuda:0")
        gen.reads(spec, P1, 0, n1, a)
        gen.reads(spec, P2, 0, n2, m1, m2)
        Lmax = 19_000
        rng = np.random.default_rng(5100)
        lens = np.clip(np.exp(rng.normal(np.log(480.0), 0.95, n3)), 200, Lmax).astype(np.int64)
        lens[:4] = (Lmax, 12_345, 200, 513)
        P3 = synthdb.read_params(spec, 5100, read_len=Lmax, sub_rate=0.075)
        rows = torch.zeros((n3, P3.row_bytes), dtype=torch.uint8, device="cuda:0")
        gen.reads(spec, P3, 0, n3, rows)
        torch.cuda.synchronize()
        singles = [bytes(r[:150]) for r in a.cpu().numpy()]
        p1 = [bytes(r[:150]) for r in m1.cpu().numpy()]
        p2 = [bytes(r[:150]) for r in m2.cpu().numpy()]
        hl = rows.cpu().numpy()
        longs = [bytes(hl[i, :int(lens[i])]) for i in range(n3)]
        e1, counts, _ = db.query(singles)
        assert np.mean(counts > 128) > 0.6, np.percentile(counts, [5, 50, 95])
        e2 = db.query(p1, p2)[0]
        e3 = db.query(longs)[0]
        db.set_lineages(spec.lineages())
        e4 = db.query(singles[:10_000], lowest=4)[0]
        e5 = db.query(longs[:200], lowest=4)[0]
        db.close()
        del a, m1, m2, rows
        torch.cuda.empty_cache()
        # one range by itself: a quarter of the locations in a table for the features that have one there
        part = api.Database.open(name, max_candidates=K, target_shard_index=2, target_shard_count=4)
        lay = part.table_layout()
        lo, hi = part.target_range()
        assert 0.2 * nloc < part.n_locations < 0.3 * nloc and 0 < lo < hi < len(spec.targets)
        assert lay["buckets"] < 0.75 * whole_layout["buckets"] and 0.15 < part.n_features / (4.0 * lay["buckets"]) < 0.35
        part.close()
        ps = api.PartSet(name, resident=4, max_candidates=K, target_shard_count=4, slot_max_queries=16384, slot_max_chars=16 << 20)
        g1 = ps.classify(singles)
        g2 = ps.classify(p1, p2)
        g3 = ps.classify(longs)
        g4 = ps.classify(singles[:10_000], lowest=4)
        g5 = ps.classify(longs[:200], lowest=4)
        ps.close()
        _same(g1, e1, "singles")
        _same(g2, e2, "pairs")
        _same(g3, e3, "long reads")
        _same(g4, e4, "singles, species level")
        _same(g5, e5, "long reads, species level")
    finally:
        for e in (".meta", ".cache0"):
            if os.path.exists(name + e):
                os.remove(name + e)
