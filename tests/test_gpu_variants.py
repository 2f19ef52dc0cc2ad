"""GPU: every shipped kernel variant of the lane path under the checkers -- the switches of mc_set_tuning (lane_fusion,
quad_lookup, gw_fuse; scale_util.VARIANTS) on the reference's golden reads (edge cases: shorter than k, tails, N, lower case, junk), on
random reads against the oracle, and with the filtered path forced onto the toy tables' lists (big_min = 0)."""
import numpy as np
import pytest

import cpuref
import scale_util
from golden.make_golden import SINGLE_RULES, PAIR_RULES
from metacache_amd import api
from test_gpu_parity import cands_equal

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("big_min", [None, 0])
@pytest.mark.parametrize("name", ["toy32", "toy16"])
def test_golden_reads_every_variant(golden, name, big_min):
    single, p1, p2 = golden.reads()
    for rname, mc, low, ins in SINGLE_RULES:
        if not mc or mc > 4:
            continue                                             # (unlimited lists: the wave kernels, not the lane path)
        db = api.Database.open(golden.db_path(name), max_candidates=mc, copy_allhits=0, slot_max_queries=700, slot_max_chars=1 << 18)
        assert db.table_layout()["location_bytes"] == 4
        if big_min is not None:
            db.set_tuning("big_min", big_min)
        exp = golden.expected(name, "single_" + rname)
        db.timing(True)
        for v in scale_util.each_variant(db):
            db.timing_reset()
            cands, counts, _ = db.query(single, lowest=low, insert_max=ins)
            ran = {k: db.timing_get(k)[1] for k in ("sketch_probe", "probe_cands", "sketch_lane", "gw_filter", "gw_filter_count")}
            fused, apart = v.startswith("lane_fusion") or v.endswith("_fused"), v.startswith("apart") or v.endswith("_apart")
            assert (ran["sketch_probe"] > 0) == fused and (ran["probe_cands"] > 0) == apart, (v, ran)
            assert db.table_layout()["direct_index"] == v.startswith("direct_index"), v
            assert (ran["gw_filter"] > 0) == (v == "apart_quad_unfused_count"), (v, ran)
            for i in range(len(single)):
                assert cands_equal(cands[i], exp[i][:mc]), (v, rname, i, cands[i], exp[i])
        db.timing(False)
        db.close()
    for rname, mc, low, ins in PAIR_RULES:
        db = api.Database.open(golden.db_path(name), max_candidates=mc, copy_allhits=0)
        if big_min is not None:
            db.set_tuning("big_min", big_min)
        exp = golden.expected(name, "pair_" + rname)
        for v in scale_util.each_variant(db):
            cands, counts, _ = db.query(p1, p2, lowest=low, insert_max=ins)
            for i in range(len(p1)):
                assert cands_equal(cands[i], exp[i]), (v, rname, i, cands[i], exp[i])
        db.close()


@pytest.mark.parametrize("big_min", [None, 0])
def test_random_reads_every_variant_against_oracle(golden, big_min):
    """junk characters, reads shorter than k, empty reads, tails of every length, odd sketching parameters"""
    rng = np.random.default_rng(23)
    odb = cpuref.oracle().open(golden.db_path("toy32"))
    single, _, _ = golden.reads()
    pool = b"".join(single[:500])
    for (s, w, st) in ((16, 127, 112), (8, 64, 49), (16, 100, 85), (12, 127, 112)):
        reads = []
        for _ in range(500):
            L = int(rng.integers(0, 520))
            o = int(rng.integers(0, len(pool) - L))
            r = bytearray(pool[o:o + L])
            for _ in range(int(rng.integers(0, 3))):
                if L:
                    r[int(rng.integers(0, L))] = int(rng.choice(list(b"NnRx-acgu")))
            reads.append(bytes(r))
        exp = [odb.query(r, b"", 4, 0, 0, sketchlen=s, winlen=w, winstride=st)[1] for r in reads]
        db = api.Database.open(golden.db_path("toy32"), max_candidates=4, copy_allhits=0, sketchlen=s, winlen=w, winstride=st)
        if big_min is not None:
            db.set_tuning("big_min", big_min)
        for v in scale_util.each_variant(db):
            cands, counts, _ = db.query(reads, lowest=0, insert_max=0)
            for i in range(len(reads)):
                assert cands_equal(cands[i], exp[i]), (v, s, w, st, i, reads[i], cands[i], exp[i])
        db.close()
    odb.close()


@pytest.mark.parametrize("align", ["0", "1"])
def test_list_alignment_on_and_off(golden, align, monkeypatch):
    """the compact store with every list on a 128-byte line of its own (mc_table_layout: list_align 32) and without: the reference's golden
    reads, singles and pairs, lane path and filtered path"""
    monkeypatch.setenv("MC_LIST_ALIGN", align)
    single, p1, p2 = golden.reads()
    for big_min in (None, 0):
        db = api.Database.open(golden.db_path("toy32"), max_candidates=2, copy_allhits=0, slot_max_queries=700, slot_max_chars=1 << 18)
        lay = db.table_layout()
        assert lay["location_bytes"] == 4 and lay["list_align"] == (32 if align == "1" else 1), lay
        if big_min is not None:
            db.set_tuning("big_min", big_min)
        exp = golden.expected("toy32", "single_c2_seq")
        cands, _, _ = db.query(single, lowest=0, insert_max=0)
        for i in range(len(single)):
            assert cands_equal(cands[i], exp[i][:2]), (align, big_min, i, cands[i], exp[i])
        exp = golden.expected("toy32", "pair_c2_seq")
        cands, _, _ = db.query(p1, p2, lowest=0, insert_max=0)
        for i in range(len(p1)):
            assert cands_equal(cands[i], exp[i]), (align, big_min, "pair", i)
        db.close()
