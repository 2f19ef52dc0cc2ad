"""GPU: ONE database key-sharded over GPUs and driven from C++ (mc_keyset_*, Mode K with 4-byte locations on the wire: every shard looks
up the features it owns for all reads, the partial lists travel as global window numbers to the shard that owns the read, the owner
runs rows 8-10 on the pieces where they lie in its receive buffer).  The result must be the single table's -- and the oracle's / the
reference's -- bit for bit.  On a one-GPU box several shards share the device (device-to-device copies stand in for the sends); the
RCCL calls themselves run with a single rank that sends to itself (MC_KEYSET_RCCL=1)."""
import os

import numpy as np
import pytest

import cpuref
from metacache_amd import api, synthdb

pytestmark = pytest.mark.gpu


def _eq(got, exp, K, tag):
    exp = exp[:K]
    for j in range(K):
        if j < len(exp):
            assert (got[j]["tgt"], got[j]["hits"], got[j]["beg"], got[j]["end"]) == (exp[j]["tgt"], exp[j]["hits"], exp[j]["beg"], exp[j]["end"]), (tag, j, got, exp)
        else:
            assert got[j]["hits"] == 0, (tag, j, got, exp)


def _same(got, exp, tag):
    bad = np.zeros(len(exp), dtype=bool)
    for f in ("tgt", "hits", "beg", "end"):
        bad |= ((got[f] != exp[f]) & ((got["hits"] > 0) | (exp["hits"] > 0))).any(axis=1)
    assert not bad.any(), (tag, int(bad.sum()), int(np.flatnonzero(bad)[0]), got[bad][:2], exp[bad][:2])


@pytest.mark.parametrize("shards,lowest,K", [(3, 0, 2), (3, 4, 3), (2, 0, 4), (5, 6, 2)])
def test_key_shards_against_the_oracle(golden, shards, lowest, K):
    single, p1, p2 = golden.reads()
    name = golden.db_path("toy32")
    odb = cpuref.oracle().open(name)
    ks = api.KeySet(name, shards=shards, max_candidates=K, slot_max_queries=700, slot_max_chars=1 << 17)     # several batches
    info = ks.info()
    whole = api.Database.open(name, max_candidates=K)
    assert info["shards"] == shards and info["devices"] == 1 and not info["rccl"] and info["locations"] == whole.n_locations
    whole.close()
    got = ks.classify(single, lowest=lowest)
    for i, s in enumerate(single):
        _, c = odb.query(s, b"", K, lowest, 0)
        _eq(got[i], c, K, ("single", i))
    gp = ks.classify(p1, p2, lowest=lowest, insert_max=700)
    for i, (a, b) in enumerate(zip(p1, p2)):
        _, c = odb.query(a, b, K, lowest, 700)
        _eq(gp[i], c, K, ("pair", i))
    info = ks.info()
    assert info["batches"] >= 4 and info["numbers_sent"] > 0
    ks.close(); odb.close()


def test_key_shard_exchange_over_rccl_single_rank(golden, monkeypatch):
    """the ncclSend / ncclRecv round of the exchange with one rank (it sends to itself) -- what several devices run between each other"""
    monkeypatch.setenv("MC_KEYSET_RCCL", "1")
    single, p1, p2 = golden.reads()
    name = golden.db_path("toy32")
    K = 2
    ks = api.KeySet(name, shards=1, max_candidates=K, slot_max_queries=900)
    assert ks.info()["rccl"]
    got = ks.classify(single[:1500])
    gp = ks.classify(p1[:300], p2[:300], insert_max=0)
    ks.close()
    whole = api.Database.open(name, max_candidates=K)
    exp, _, _ = whole.query(single[:1500])
    expp, _, _ = whole.query(p1[:300], p2[:300])
    whole.close()
    _same(got, exp, "singles")
    _same(gp, expp, "pairs")


def test_key_shards_at_midscale_filtered_regime():
    """A 15 Gbp cut of the bench collection (195 locations per 150 bp read: the filtered path's regime) written as database files, then
    opened as 4 key shards: singles, pairs and long reads (window ranges up to 171, sorted lists) against the single table holding
    everything -- whose results tests/test_gpu_reference_midscale.py holds against the reference itself on the same collection."""
    import torch
    shm = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
    name = os.path.join(shm, f"mc_keyset_{os.getpid()}")
    spec = synthdb.phylogeny(200, 4, 5, 2_500_000, 5_000_000, seed=3100)
    K = 2
    n1, n2, n3 = 40_000, 8_000, 600
    try:
        db, _ = synthdb.build_database(spec, shards=2, max_candidates=K, write_to=name)
        assert db.table_layout()["location_bytes"] == 4
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
        e4 = db.query(singles[:10_000], lowest=4)[0]               # taxon merging (per-number taxon lookups in the counting kernel, struck by taxon)
        e5 = db.query(longs[:200], lowest=4)[0]
        db.close()
        del a, m1, m2, rows
        torch.cuda.empty_cache()
        ks = api.KeySet(name, shards=4, max_candidates=K, slot_max_queries=16384, slot_max_chars=16 << 20)
        assert ks.info()["locations"] > 2_000_000_000
        g1 = ks.classify(singles)
        g1info = ks.info()
        assert g1info["reads_filtered"] > 0.6 * n1                # the owners' filtered path (gw_filter / gw_count on the receive buffer)
        g2 = ks.classify(p1, p2)
        g3 = ks.classify(longs)
        g4 = ks.classify(singles[:10_000], lowest=4)
        g5 = ks.classify(longs[:200], lowest=4)
        info = ks.info()
        sent = info["numbers_sent"]
        assert info["reads_filtered"] > g1info["reads_filtered"] + n2 // 2 + n3 // 2
        ks.close()
        _same(g1, e1, "singles")
        _same(g2, e2, "pairs")
        _same(g3, e3, "long reads")
        _same(g4, e4, "singles, species level")
        _same(g5, e5, "long reads, species level")
        # every location of every read crossed the exchange exactly once, 4 bytes each
        assert sent > int(counts.sum())
    finally:
        for e in (".meta", ".cache0"):
            if os.path.exists(name + e):
                os.remove(name + e)


def test_key_shards_need_the_window_numbering(golden, monkeypatch):
    """a table that keeps 8-byte locations has nothing 4 bytes wide to send: mc_keyset_open says so instead of failing at the first batch"""
    monkeypatch.setenv("MC_COMPACT_LOCATIONS", "0")
    with pytest.raises(api.McError, match="global window numbers"):
        api.KeySet(golden.db_path("toy32"), shards=2, max_candidates=2)
