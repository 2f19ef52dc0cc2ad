"""CPU: pins oracle/mc_oracle.c (our C restatement) against the golden vectors the reference produced
(tests/golden/, see make_golden.py) and -- where oracle/_ref is present -- live against the reference."""
import numpy as np
import pytest

import cpuref
from conftest import split
from golden.make_golden import SINGLE_RULES, PAIR_RULES


@pytest.fixture(scope="module")
def orc():
    return cpuref.oracle()


def test_sketch_vectors(orc, golden):
    z = golden.npz("sketch_vectors")
    strings = [x.tobytes() for x in split(z["strings"], z["strings_off"])]
    for pi, (k, s, w, st) in enumerate(z["params"]):
        feats, counts, nwin = z[f"p{pi}_feats"], z[f"p{pi}_counts"], z[f"p{pi}_nwin"]
        fo = co = 0
        for x, nw in zip(strings, nwin):
            f, c = orc.sketch(x, int(k), int(s), int(w), int(st))
            assert len(c) == nw
            assert np.array_equal(c, counts[co:co + nw])
            assert np.array_equal(f.reshape(-1), feats[fo:fo + nw * s])
            fo += nw * int(s); co += nw


def _check_db(orc, golden, name, tb):
    single, p1, p2 = golden.reads()
    db = orc.open(golden.db_path(name))
    z = golden.npz(name + "_expected")
    assert db.ref.lib.mco_db_target_id_bytes(db.h) == tb
    assert db.info() == list(map(int, z["info"]))
    assert np.array_equal(db.lineages(), z["lineages"])
    assert [db.target_name(t) for t in range(db.n_targets)] == list(z["target_names"])
    exp_hits = golden.expected(name, "single_allhits")
    exp = {r[0]: golden.expected(name, "single_" + r[0]) for r in SINGLE_RULES}
    for i, s in enumerate(single):
        for rname, mc, low, ins in SINGLE_RULES:
            h, c = db.query(s, b"", mc, low, ins)
            assert np.array_equal(h, exp_hits[i]), (i, rname)
            assert np.array_equal(c, exp[rname][i]), (i, rname)
    exp_hits = golden.expected(name, "pair_allhits")
    exp = {r[0]: golden.expected(name, "pair_" + r[0]) for r in PAIR_RULES}
    for i, (a, b) in enumerate(zip(p1, p2)):
        for rname, mc, low, ins in PAIR_RULES:
            h, c = db.query(a, b, mc, low, ins)
            assert np.array_equal(h, exp_hits[i]), (i, rname)
            assert np.array_equal(c, exp[rname][i]), (i, rname)
    # row 13 modifiers
    idx = z["maxloc2_idx"]
    db.set_max_locations_per_feature(2)
    eh, ec = golden.expected(name, "maxloc2_allhits"), golden.expected(name, "maxloc2_c2_seq")
    for j, i in enumerate(idx):
        h, c = db.query(single[i], b"", 2, 0, 0)
        assert np.array_equal(h, eh[j]) and np.array_equal(c, ec[j]), i
    db.close()
    db = orc.open(golden.db_path(name))
    assert db.remove_features_with_more_locations_than(3) == int(z["rmover_removed"][0])
    eh, ec = golden.expected(name, "rmover_allhits"), golden.expected(name, "rmover_c2_seq")
    for j, i in enumerate(idx):
        h, c = db.query(single[i], b"", 2, 0, 0)
        assert np.array_equal(h, eh[j]) and np.array_equal(c, ec[j]), i
    db.close()


def test_query_toy32(orc, golden):
    _check_db(orc, golden, "toy32", 4)


def test_query_toy16(orc, golden):
    _check_db(orc, golden, "toy16", 2)


def test_query_multipart_reference_quirk(orc, golden):
    """2-part DB, mode 0 = the reference's actual (history dependent) behaviour, reads in order."""
    single, _, _ = golden.reads()
    db = orc.open(golden.db_path("toy32p2"))
    z = golden.npz("toy32p2_expected")
    assert db.info() == list(map(int, z["info"]))
    eh, ec = golden.expected("toy32p2", "single_allhits"), golden.expected("toy32p2", "single_c2_seq")
    for i, s in enumerate(single):
        h, c = db.query(s, b"", 2, 0, 0, mode=0)
        assert np.array_equal(h, eh[i]), i
        assert np.array_equal(c, ec[i]), i
    db.close()


def test_multipart_intended_equals_single_part_hits(orc, golden):
    """mode 1 (intended semantics): the multiset of hits over both parts equals the 1-part DB's
    (the part split only changes ordering) -- target ids differ between the two builds, so compare
    through target names."""
    single, _, _ = golden.reads()
    d1 = orc.open(golden.db_path("toy32"))
    d2 = orc.open(golden.db_path("toy32p2"))
    n1 = [d1.target_name(t) for t in range(d1.n_targets)]
    n2 = [d2.target_name(t) for t in range(d2.n_targets)]
    checked = 0
    for s in single[:300]:
        # buckets at the 254-location cap are truncated per part, so they legitimately differ
        feats, _ = orc.sketch(s, d1.k, d1.s, d1.w, d1.stride)
        fl = [int(f) for f in feats.reshape(-1) if f != 0xFFFFFFFF]
        if any(len(d1.lookup(f)) >= 254 or len(d2.lookup(f, 0)) >= 254 or len(d2.lookup(f, 1)) >= 254 for f in fl):
            continue
        checked += 1
        h1, _ = d1.query(s, mode=1)
        h2, _ = d2.query(s, mode=1)
        a = sorted((n1[t], int(w)) for w, t in zip(h1["win"], h1["tgt"]))
        b = sorted((n2[t], int(w)) for w, t in zip(h2["win"], h2["tgt"]))
        assert a == b
    assert checked > 150
    d1.close(); d2.close()


@pytest.mark.skipif(not cpuref.have_reference(4), reason="oracle/_ref not built")
def test_live_against_reference(orc, golden):
    """Fresh random reads + odd sketching parameters, oracle vs the real reference in-process."""
    rng = np.random.default_rng(7)
    ref = cpuref.reference(4)
    rdb = ref.open(golden.db_path("toy32"))
    odb = orc.open(golden.db_path("toy32"))
    single, _, _ = golden.reads()
    pool = b"".join(single[:400])
    for it in range(600):
        L = int(rng.integers(0, 700))
        st = int(rng.integers(0, len(pool) - L))
        s = bytearray(pool[st:st + L])
        for _ in range(int(rng.integers(0, 4))):
            if L:
                s[int(rng.integers(0, L))] = int(rng.choice(list(b"NnRxacgtu-")))
        mc = int(rng.integers(0, 5)); low = int(rng.choice([0, 0, 4, 6, 10])); ins = int(rng.choice([0, 0, 300, 1000]))
        kw = {}
        if it % 3 == 0:
            kw = dict(sketchlen=int(rng.choice([4, 16, 24])), winlen=int(rng.choice([64, 127, 200])),
                      winstride=int(rng.choice([30, 112, 150])))
        a = rdb.query(bytes(s), b"", mc, low, ins, **kw)
        b = odb.query(bytes(s), b"", mc, low, ins, **kw)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), (it, L, mc, low, ins, kw)
        f1 = ref.sketch(bytes(s), 16, 16, 127, 112); f2 = orc.sketch(bytes(s), 16, 16, 127, 112)
        assert np.array_equal(f1[0], f2[0]) and np.array_equal(f1[1], f2[1])
    rdb.close(); odb.close()


def test_bulk_pair_entry_equals_the_goldens(orc, golden):
    """mco_query_many_pairs (the threaded bulk entry bench.py's --pairs baseline runs through) against the reference's pair goldens"""
    _, p1, p2 = golden.reads()
    db = orc.open(golden.db_path("toy32"))
    s1 = np.frombuffer(b"".join(p1), dtype=np.uint8); s2 = np.frombuffer(b"".join(p2), dtype=np.uint8)
    o1 = np.zeros(len(p1) + 1, np.uint64); o1[1:] = np.cumsum([len(x) for x in p1])
    o2 = np.zeros(len(p2) + 1, np.uint64); o2[1:] = np.cumsum([len(x) for x in p2])
    for rname, mc, low, ins in PAIR_RULES:
        exp = golden.expected("toy32", "pair_" + rname)
        for threads in (1, 3):
            _, c = db.query_many_pairs(s1, o1, s2, o2, max_cand=mc, lowest=low, insert_max=ins, threads=threads)
            for i in range(len(p1)):
                e = exp[i]
                assert int((c[i]["hits"] > 0).sum()) == len(e), (rname, i)
                for j in range(len(e)):
                    assert tuple(c[i][j][f] for f in ("tgt", "hits", "beg", "end")) == tuple(e[j][f] for f in ("tgt", "hits", "beg", "end")), (rname, i, j)
    db.close()
