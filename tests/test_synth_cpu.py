"""CPU: the synthetic-workload function (metacache_amd/synth) and the oracle's restatement of the database BUILD, which the
RefSeq-scale parity checks stand on (the checker builds the buckets of a read sample's features itself)."""
import ctypes as C

import numpy as np

import cpuref
import scale_util
from metacache_amd import synth, synthdb


def test_synthetic_phylogeny_is_a_function_with_the_stated_divergences():
    spec = synthdb.phylogeny(3, 2, 3, 20_000, 40_000, seed=7)
    cs = synthdb.CpuSynth()
    g0, g1, g3, g6 = (cs.target(spec, t) for t in (0, 1, 3, 6))
    assert np.array_equal(g0, cs.target(spec, 0))                                  # deterministic
    assert np.array_equal(g0[1000:1500], cs.target(spec, 0, 1000, 500))            # any stretch on its own
    T = spec.targets
    assert abs((g0 != g1).mean() - T["thr_strain"][1] / 2 ** 32) < 0.003           # strain vs its species sequence (strain 0)
    assert 0.03 < (g0 != g3).mean() < 0.25                                         # two species of a genus
    n = min(len(g0), len(g6))
    assert 0.70 < (g0[:n] != g6[:n]).mean() < 0.80                                 # unrelated genera
    assert set(np.unique(g0)) <= set(b"ACGT")
    P = synthdb.read_params(spec, 11)
    reads, org = cs.reads(spec, P, 0, 200), cs.origins(spec, P, 0, 200)
    assert np.array_equal(reads[5:9], cs.reads(spec, P, 5, 4))                     # read i does not depend on the batch
    diffs = 0
    for i in range(200):
        t, st, rev, _ = map(int, org[i])
        g = cs.target(spec, t, st, 150)
        ref = synth.revcomp(g) if rev else g
        diffs += int((reads[i, :150] != ref).sum())
        assert not reads[i, 150:].any()
    assert 0.002 < diffs / (200 * 150) < 0.03                                      # 1 % substitutions + 0.1 % N
    P2 = synthdb.read_params(spec, 12, paired=True)
    a, b = cs.reads(spec, P2, 0, 50)
    org = cs.origins(spec, P2, 0, 50)
    for i in range(50):
        t, st, rev, fr = map(int, org[i])
        frag = cs.target(spec, t, st, fr)
        m1, m2 = frag[:150], synth.revcomp(frag)[:150]
        if rev:
            m1, m2 = m2, m1
        assert (a[i, :150] != m1).sum() <= 8 and (b[i, :150] != m2).sum() <= 8


def _toy_genomes_in_target_order(golden):
    from golden import make_golden
    rng = np.random.default_rng(20240928)
    genomes, headers, _ = make_golden.make_genomes(rng)
    ref = cpuref.oracle().open(golden.db_path("toy32"))
    acc = [h.split()[0] for h in headers]
    order = [acc.index(ref.target_name(t)) for t in range(ref.n_targets)]
    return ref, [genomes[i] for i in order]


def test_oracle_build_restatement_against_the_reference_built_database(golden):
    """mco_db_build (the oracle's restatement of database::add_target / host_hashmap::add_target) over the genomes the reference's
    own `build` was given must reproduce every bucket of the reference-built toy32 files.  One documented difference: the
    reference's multi-threaded build merges per-thread tables with a range insert that caps buckets at max_bucket_size() = 255
    (hash_multimap.hpp:304-306), the single-table insertion path modelled here (and by the GPU builder) caps at 254
    (host_hashmap.hpp:593-605) -- those buckets must agree in their first 254 locations."""
    ref, gs = _toy_genomes_in_target_order(golden)
    GEN = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32, C.c_void_p)
    cb = GEN(lambda user, t, dst: C.memmove(dst, gs[t].ctypes.data, gs[t].size))
    lengths = np.array([g.size for g in gs], dtype=np.uint32)
    k1, s1, o1, v1 = ref.part_arrays()
    for threads in (1, 3):
        db = cpuref.oracle().build_db(lengths, C.cast(cb, C.c_void_p).value, None, threads=threads)
        k2, s2, _, _ = db.part_arrays()
        assert len(k1) == len(k2) and set(k1.tolist()) == set(k2.tolist())
        capped = 0
        for i in range(len(k1)):
            got = db.lookup(int(k1[i]))
            exp = v1[int(o1[i]):int(o1[i]) + int(s1[i])]
            if s1[i] == 255:
                capped += 1
                assert len(got) == 254 and np.array_equal(got, exp[:254])
            else:
                assert np.array_equal(got, exp), (i, k1[i])
        assert capped > 10                                                           # the cap was exercised
        assert np.array_equal(db.target_windows, [synth_windows(len(g)) for g in gs])
        db.close()
    # restricted to some features: exactly those buckets, identical
    want = k1[::7].copy()
    db = cpuref.oracle().build_db(lengths, C.cast(cb, C.c_void_p).value, None, wanted=want, threads=2)
    k2, _, _, _ = db.part_arrays()
    assert sorted(k2.tolist()) == sorted(want.tolist())
    for f in want[:500]:
        exp = ref.lookup(int(f))[:254]
        assert np.array_equal(db.lookup(int(f)), exp)
    db.close(); ref.close()


def synth_windows(L, k=16, w=127, stride=112):
    if L <= w:
        return 1 if L >= k else 0
    nf = (L - w) // stride + 1
    return nf + (1 if L - nf * stride >= k else 0)


def test_oracle_query_many_threads_equal_single_thread(golden):
    single, _, _ = golden.reads()
    reads = [r for r in single[:600] if len(r) > 0]
    seqs = np.frombuffer(b"".join(reads), dtype=np.uint8)
    offs = np.zeros(len(reads) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(r) for r in reads])
    db = cpuref.oracle().open(golden.db_path("toy32"))
    _, c1 = db.query_many(seqs, offs, max_cand=3, threads=1)
    _, c4 = db.query_many(seqs, offs, max_cand=3, threads=4)
    assert np.array_equal(c1, c4)
    for i in (0, 17, 333):
        _, e = db.query(reads[i], b"", 3, 0, 0)
        assert [tuple(x) for x in c1[i][:len(e)].tolist()] == [tuple(x) for x in e.tolist()]
    db.close()


def test_oracle_database_from_synthetic_collection_matches_python_restatement():
    """synthdb.oracle_database = mco_db_build fed by the C generator: a handful of buckets against a pure-Python walk."""
    spec = synthdb.phylogeny(2, 2, 2, 3_000, 5_000, seed=3)
    cs = synthdb.CpuSynth()
    odb = scale_util.oracle_database(spec, None, threads=2)
    expect = {}
    for t in range(len(spec.targets)):
        g = cs.target(spec, t).tobytes()
        feats, counts = cpuref.oracle().sketch(g)
        for w in range(len(counts)):
            for f in feats[w, :counts[w]]:
                expect.setdefault(int(f), []).append((t << 32) | w)
    keys, sizes, _, _ = odb.part_arrays()
    assert len(keys) == len(expect)
    for f in list(expect)[::37]:
        assert odb.lookup(f).tolist() == expect[f][:254]
    odb.close()


def test_oracle_fast_window_sketcher_equals_the_plain_one():
    """mco_db_build sketches with a rolling form of the window sketcher; it must give what the plain restatement (pinned against the
    reference's vectors in test_oracle_golden.py) gives: random windows with ambiguity codes, lower case, U, all lengths around k."""
    lib = cpuref.oracle().lib
    for f in (lib.mco_sketch_window_plain, lib.mco_sketch_window_fast):
        f.argtypes = [C.c_void_p, C.c_uint64, C.c_uint, C.c_uint32, C.c_void_p]
        f.restype = C.c_int
    rng = np.random.default_rng(1)
    alphabet = np.frombuffer(b"ACGTacgtNUuR-", dtype=np.uint8)
    p = np.array([.22, .22, .22, .22, .02, .02, .02, .02, .01, .01, .005, .01, .005])
    for it in range(6000):
        n = int(rng.integers(0, 200)); k = int(rng.choice([16, 16, 16, 10, 7, 12, 1])); s = int(rng.choice([16, 16, 8, 4, 32, 1]))
        seq = rng.choice(alphabet, size=n + 1, p=p / p.sum())
        if it % 50 == 0 and n >= 40:
            seq[10:40] = seq[60:90] if n >= 90 else seq[10]                      # repeats: equal hashes inside a window
        a = np.zeros(64, np.uint32); b = np.zeros(64, np.uint32)
        ca = lib.mco_sketch_window_plain(seq.ctypes.data, n, k, s, a.ctypes.data)
        cb = lib.mco_sketch_window_fast(seq.ctypes.data, n, k, s, b.ctypes.data)
        assert ca == cb and (ca <= 0 or np.array_equal(a[:ca], b[:cb])), (it, n, k, s)
