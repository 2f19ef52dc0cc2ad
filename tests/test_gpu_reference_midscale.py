"""GPU: the REFERENCE itself (oracle/_ref: muellan/metacache's own sources compiled by oracle/Makefile) as the witness of the filtered
candidate path -- the kernels that carry the headline (gw_filter_kernel, gw_count_kernel; big_filter_kernel, big_count_kernel on the
8-byte store).  A 15 Gbp cut of the bench collection (4 000 targets, 2.2 x 10^9 locations, 195 locations per 150 bp read) is built on
the GPU in key shards and written as database files by the streaming writer; the reference loads those files (database.hpp:399-407
query_host on ITS hash table, not the oracle's restatement of it) and every candidate of 60 000 single reads and 10 000 read pairs is
compared with the GPU's -- both location stores, and once more with every list above 64 locations forced through the filter."""
import os

import numpy as np
import pytest

import cpuref
import scale_util
from metacache_amd import api, synthdb

pytestmark = pytest.mark.gpu
K = 2


def _same(got, exp, tag):
    bad = np.zeros(len(exp), dtype=bool)
    for f in ("tgt", "hits", "beg", "end"):
        bad |= ((got[f] != exp[f]) & ((got["hits"] > 0) | (exp["hits"] > 0))).any(axis=1)
    assert not bad.any(), (tag, int(bad.sum()), int(np.flatnonzero(bad)[0]), got[bad][:2], exp[bad][:2])


@pytest.mark.skipif(not cpuref.have_reference(4), reason="oracle/_ref not built (needs /root/reference at build time)")
def test_reference_witnesses_the_filtered_path(monkeypatch):
    import torch
    shm = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
    name = os.path.join(shm, f"mc_refmid_{os.getpid()}")
    spec = synthdb.phylogeny(200, 4, 5, 2_500_000, 5_000_000, seed=3100)          # bench.CFG2 at scale 0.1
    n1, n2 = 60_000, 10_000
    try:
        monkeypatch.setenv("MC_COMPACT_LOCATIONS", "1")
        db, info = synthdb.build_database(spec, shards=2, max_candidates=K, write_to=name)
        assert db.table_layout()["location_bytes"] == 4 and os.path.getsize(name + ".cache0") > 15e9
        gen = synthdb.GpuSynth(0)
        P1 = synthdb.read_params(spec, 3100)
        P2 = synthdb.read_params(spec, 4100, paired=True)
        a = torch.zeros((n1, P1.row_bytes), dtype=torch.uint8, device="cuda:0")
        m1 = torch.zeros((n2, P2.row_bytes), dtype=torch.uint8, device="cuda:0")
        m2 = torch.zeros((n2, P2.row_bytes), dtype=torch.uint8, device="cuda:0")
        gen.reads(spec, P1, 0, n1, a)
        gen.reads(spec, P2, 0, n2, m1, m2)
        torch.cuda.synchronize()
        ah = a.cpu().numpy()
        singles = [bytes(r[:150]) for r in ah]
        p1 = [bytes(r[:150]) for r in m1.cpu().numpy()]
        p2 = [bytes(r[:150]) for r in m2.cpu().numpy()]
        got = {}
        c, counts, _ = db.query(singles)
        assert np.mean(counts > 128) > 0.6, np.percentile(counts, [5, 50, 95])      # the filtered path's regime
        st = db.last_batch_stats()
        got["compact"] = (c, db.query(p1, p2)[0])
        # every shipped variant of the lane path's kernels on the reference's reads (scale_util.VARIANTS)
        for v in scale_util.each_variant(db):
            got["compact, " + v] = (db.query(singles)[0], db.query(p1, p2)[0])
        db.set_tuning("big_min", 0)
        got["compact, big_min 0"] = (db.query(singles)[0], db.query(p1, p2)[0])
        for v in scale_util.each_variant(db, ("apart_quad_unfused_count",)):
            got["compact, big_min 0, " + v] = (db.query(singles)[0], db.query(p1, p2)[0])
        db.close()
        monkeypatch.setenv("MC_COMPACT_LOCATIONS", "0")
        db, _ = synthdb.build_database(spec, shards=2, max_candidates=K)
        assert db.table_layout()["location_bytes"] == 8
        got["wide"] = (db.query(singles)[0], db.query(p1, p2)[0])
        db.close()
        # ---- the reference on the files
        threads = max(4, 2 * scale_util.effective_cpus())
        ref = cpuref.reference(4).open(name)
        seqs = np.ascontiguousarray(ah[:, :150]).reshape(-1)
        offs = np.arange(n1 + 1, dtype=np.uint64) * np.uint64(150)
        _, exp1 = ref.query_many(seqs, offs, max_cand=K, lowest=0, insert_max=0, threads=threads)
        exp2 = np.zeros((n2, K), dtype=exp1.dtype)
        for i in range(n2):
            _, e = ref.query(p1[i], p2[i], K, 0, 0)
            for j in range(min(K, len(e))):
                for f in ("tgt", "hits", "beg", "end"):
                    exp2[i, j][f] = e[j][f]
        ref.close()
        for tag, (g1, g2) in got.items():
            _same(g1, exp1, (tag, "singles"))
            _same(g2, exp2, (tag, "pairs"))
    finally:
        for e in (".meta", ".cache0"):
            if os.path.exists(name + e):
                os.remove(name + e)
