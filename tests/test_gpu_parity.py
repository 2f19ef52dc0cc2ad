"""GPU: the HIP path (through the C ABI) against the golden vectors of the reference and against the
C oracle on the same seeded inputs.  Bit-exact: every field of every candidate and every location."""
import numpy as np
import pytest

import cpuref
from golden.make_golden import SINGLE_RULES, PAIR_RULES
from metacache_amd import api

pytestmark = pytest.mark.gpu


def cands_equal(dev_row, exp):
    """dev_row: [K] cand_dtype (unused = hits 0); exp: reference candidates of one query"""
    n = len(exp)
    used = int((dev_row["hits"] > 0).sum())
    if used != n:
        return False
    d = dev_row[:n]
    return (np.array_equal(d["tgt"], exp["tgt"]) and np.array_equal(d["hits"], exp["hits"]) and
            np.array_equal(d["beg"], exp["beg"]) and np.array_equal(d["end"], exp["end"]))


def taxid_of(db_taxa, lin, tgt, lowest):
    if lowest == 0:
        return -int(tgt) - 1
    for r in range(lowest, api.NUM_RANKS):
        if lin[tgt, r]:
            return db_taxa[lin[tgt, r] - 1][0]
    return 0


@pytest.mark.parametrize("store", ["compact", "wide"])
@pytest.mark.parametrize("name", ["toy32", "toy16"])
def test_golden_single_and_pairs(golden, name, store, monkeypatch):
    # both location stores: 4 bytes per location (mc_open_database takes the range from the reference-built file's target metadata)
    # and the reference's 8
    monkeypatch.setenv("MC_COMPACT_LOCATIONS", "1" if store == "compact" else "0")
    single, p1, p2 = golden.reads()
    exp_hits = golden.expected(name, "single_allhits")
    for rname, mc, low, ins in SINGLE_RULES:
        K = mc if mc else 64
        db = api.Database.open(golden.db_path(name), max_candidates=K, copy_allhits=1, slot_max_queries=700, slot_max_chars=1 << 18)
        assert db.table_layout()["location_bytes"] == (4 if store == "compact" else 8)
        cands, counts, allhits = db.query(single, lowest=low, insert_max=ins)
        exp = golden.expected(name, "single_" + rname)
        taxa, lin = db.taxa(), db.lineages()
        for i in range(len(single)):
            assert counts[i] == len(exp_hits[i]), (rname, i)
            assert np.array_equal(allhits[i]["win"], exp_hits[i]["win"]) and np.array_equal(allhits[i]["tgt"], exp_hits[i]["tgt"]), (rname, i)
            e = exp[i][:K]
            assert cands_equal(cands[i], e), (rname, i, cands[i], e)
            for c in e:
                assert taxid_of(taxa, lin, c["tgt"], low) == c["taxid"]
        db.close()
    exp_hits = golden.expected(name, "pair_allhits")
    for rname, mc, low, ins in PAIR_RULES:
        db = api.Database.open(golden.db_path(name), max_candidates=mc, copy_allhits=1)
        cands, counts, allhits = db.query(p1, p2, lowest=low, insert_max=ins)
        exp = golden.expected(name, "pair_" + rname)
        for i in range(len(p1)):
            assert np.array_equal(allhits[i]["win"], exp_hits[i]["win"]) and np.array_equal(allhits[i]["tgt"], exp_hits[i]["tgt"]), (rname, i)
            assert cands_equal(cands[i], exp[i]), (rname, i, cands[i], exp[i])
        db.close()


@pytest.mark.parametrize("name", ["toy32", "toy16"])
def test_golden_load_time_modifiers(golden, name):
    single, _, _ = golden.reads()
    z = golden.npz(name + "_expected")
    idx = z["maxloc2_idx"]
    sub = [single[i] for i in idx]
    for key, kw in (("maxloc2", dict(max_locations_per_feature=2)), ("rmover", dict(remove_overpopulated=3))):
        db = api.Database.open(golden.db_path(name), max_candidates=2, copy_allhits=1, **kw)
        cands, counts, allhits = db.query(sub)
        eh, ec = golden.expected(name, key + "_allhits"), golden.expected(name, key + "_c2_seq")
        for j in range(len(sub)):
            assert np.array_equal(allhits[j]["win"], eh[j]["win"]) and np.array_equal(allhits[j]["tgt"], eh[j]["tgt"]), (key, j)
            assert cands_equal(cands[j], ec[j]), (key, j)
        db.close()


def test_candidates_without_allhits_and_small_slots(golden):
    """allhits off (location lists stay in LDS), tiny slots (many batches), odd K."""
    single, _, _ = golden.reads()
    db = api.Database.open(golden.db_path("toy32"), max_candidates=3, copy_allhits=0, slot_max_queries=37, slot_max_chars=1 << 14)
    cands, counts, _ = db.query(single[:600], lowest=4)
    exp = golden.expected("toy32", "single_c3_species")
    eh = golden.expected("toy32", "single_allhits")
    for i in range(600):
        assert counts[i] == len(eh[i])
        assert cands_equal(cands[i], exp[i]), i
    db.close()


def test_random_reads_against_oracle(golden):
    """fresh seeded reads incl. odd query sketching parameters: HIP path vs the C oracle"""
    rng = np.random.default_rng(11)
    orc = cpuref.oracle()
    odb = orc.open(golden.db_path("toy32"))
    single, _, _ = golden.reads()
    pool = b"".join(single[:500])
    for (s, w, st) in ((16, 127, 112), (8, 64, 49), (32, 300, 290), (16, 127, 40), (16, 100, 130)):
        reads = []
        for _ in range(400):
            L = int(rng.integers(0, 900))
            o = int(rng.integers(0, len(pool) - L))
            r = bytearray(pool[o:o + L])
            for _ in range(int(rng.integers(0, 3))):
                if L:
                    r[int(rng.integers(0, L))] = int(rng.choice(list(b"NnRx-acgu")))
            reads.append(bytes(r))
        db = api.Database.open(golden.db_path("toy32"), max_candidates=4, copy_allhits=1, sketchlen=s, winlen=w, winstride=st)
        cands, counts, allhits = db.query(reads, lowest=0, insert_max=0)
        for i, r in enumerate(reads):
            h, c = odb.query(r, b"", 4, 0, 0, sketchlen=s, winlen=w, winstride=st)
            assert np.array_equal(allhits[i]["win"], h["win"]) and np.array_equal(allhits[i]["tgt"], h["tgt"]), (s, w, st, i)
            assert cands_equal(cands[i], c), (s, w, st, i)
        db.close()
    odb.close()


def test_device_path_features_match_oracle(golden):
    """window sketches straight from the device buffers vs the oracle's sketcher"""
    import torch
    single, _, _ = golden.reads()
    reads = single[1590:1700]
    orc = cpuref.oracle()
    db = api.Database.open(golden.db_path("toy32"), max_candidates=2)
    offs, chunks, pos = [], [], 0
    for r in reads:
        offs.append((pos, len(r), pos, 0))
        pad = (-len(r)) % 4
        chunks.append(r + b"\0" * pad)
        pos += len(r) + pad
    buf = np.frombuffer(b"".join(chunks) + b"\0" * 16, dtype=np.uint8)
    dseq = torch.from_numpy(buf.copy()).cuda()
    dq = torch.from_numpy(np.array(offs, dtype=np.uint32).view(np.int32).reshape(-1)).cuda()
    res = db.query_device(dseq.data_ptr(), dq.data_ptr(), len(reads), pos, max_win_uniform=3, want_features=True)
    db.synchronize()
    n = len(reads)
    import ctypes as C
    winoff = torch.empty(n + 1, dtype=torch.int32, device="cuda")
    C.cdll.LoadLibrary("libamdhip64.so").hipMemcpy(C.c_void_p(winoff.data_ptr()), C.c_void_p(res.win_offsets), (n + 1) * 4, 3)
    wo = winoff.cpu().numpy().astype(np.int64)
    feats = torch.empty(int(wo[-1]) * 16, dtype=torch.int32, device="cuda")
    C.cdll.LoadLibrary("libamdhip64.so").hipMemcpy(C.c_void_p(feats.data_ptr()), C.c_void_p(res.features), int(wo[-1]) * 64, 3)
    f = feats.cpu().numpy().view(np.uint32).reshape(-1, 16)
    for i, r in enumerate(reads):
        ef, ec = orc.sketch(r, 16, 16, 127, 112)
        assert wo[i + 1] - wo[i] == len(ec), i
        assert np.array_equal(f[wo[i]:wo[i + 1]], ef), i
    st = db.last_batch_stats()
    assert st["windows"] == wo[-1]
    db.close()


def test_many_reads_random_db_lane_and_wave_paths_against_oracle(tmp_path):
    """A database of unrelated random genomes produces spurious single hits on other targets, i.e. many
    equal-hit ties in the top-2 list -- the situation in which candidate ordering bugs show.  Both the
    lane-parallel short-read path and the wave-per-query path must agree with the oracle."""
    import os
    from metacache_amd import synth
    rng = np.random.default_rng(99)
    genomes = [synth.random_genome(rng, 300_000) for _ in range(16)]
    bld = api.Builder(target_id_bytes=4, max_candidates=2)
    for i, g in enumerate(genomes):
        bld.add_target(g, f"R{i:04d}.1", parent_taxid=1000 + i)
    name = str(tmp_path / "rand16")
    bld.finish(load=False)
    bld.write(name, [(1, 1, 20, "root")] + [(1000 + i, 1, 4, f"sp{i}") for i in range(16)])
    bld.free()
    reads, _, _ = synth.sample_reads(rng, genomes, 30000, 150, 0.01, 0.002)
    reads = [bytes(r) for r in reads]
    odb = cpuref.oracle().open(name)
    seqs = np.frombuffer(b"".join(reads), dtype=np.uint8)
    offs = np.arange(len(reads) + 1, dtype=np.uint64) * np.uint64(150)
    _, exp = odb.query_many(seqs, offs, max_cand=2)
    odb.close()
    ties = int(((exp["hits"][:, 0] > 1) & (exp["hits"][:, 1] == 1)).sum())
    assert ties > 50                                   # the interesting situation really occurs
    for env in ("0", "1"):
        os.environ["MC_NO_LANE_PATH"] = env
        try:
            db = api.Database.open(name, max_candidates=2, slot_max_queries=1 << 15, slot_max_chars=1 << 23)
            cands, counts, _ = db.query(reads)
            db.close()
        finally:
            os.environ.pop("MC_NO_LANE_PATH", None)
        for f in ("hits", "beg", "end"):
            assert np.array_equal(cands[f], exp[f]), (env, f)
        used = exp["hits"] > 0
        assert np.array_equal(cands["tgt"][used], exp["tgt"][used]), env


@pytest.mark.parametrize("lowest,K", [(0, 2), (4, 3)])
def test_two_part_database_intended_semantics(golden, lowest, K):
    """Database written by the reference with -parts 2: all parts live in ONE device table, results are the
    intended 'per-part sorted lists concatenated in part order' (oracle mode 1; the in-process reference
    itself is history dependent for P > 1)."""
    single, p1, p2 = golden.reads()
    odb = cpuref.oracle().open(golden.db_path("toy32p2"))
    for allhits_flag in (1, 0):
        db = api.Database.open(golden.db_path("toy32p2"), max_candidates=K, copy_allhits=allhits_flag)
        assert db.n_parts == 2 and db.n_locations == odb.n_locations
        cands, counts, allhits = db.query(single, lowest=lowest)
        for i, s in enumerate(single):
            h, c = odb.query(s, b"", K, lowest, 0, mode=1)
            assert counts[i] == len(h), i
            if allhits_flag:
                assert np.array_equal(allhits[i]["win"], h["win"]) and np.array_equal(allhits[i]["tgt"], h["tgt"]), i
            assert cands_equal(cands[i], c), (i, cands[i], c)
        cands, counts, allhits = db.query(p1, p2, lowest=lowest)
        for i, (a, b) in enumerate(zip(p1, p2)):
            h, c = odb.query(a, b, K, lowest, 0, mode=1)
            assert cands_equal(cands[i], c), i
        db.close()
    odb.close()


def test_mode_p_single_parts_merge_equals_whole_database(golden):
    """one part per context (what one GPU holds in Mode P) + merge_part_candidates == all parts in one context"""
    import torch
    from metacache_amd.distributed import merge_part_candidates
    single, _, _ = golden.reads()
    reads = single[:700]
    K = 2
    whole = api.Database.open(golden.db_path("toy32p2"), max_candidates=K)
    cw, _, _ = whole.query(reads)
    whole.close()
    per_part = []
    for p in range(2):
        db = api.Database.open(golden.db_path("toy32p2"), max_candidates=K, single_part=p)
        assert db.n_parts == 1
        c, _, _ = db.query(reads)
        db.close()
        t = np.stack([c["tgt"], c["hits"], c["beg"], c["end"]], axis=-1).astype(np.uint32).view(np.int32)
        per_part.append(torch.from_numpy(t.copy()))
    merged = merge_part_candidates(per_part).numpy().view(np.uint32)
    for f, j in (("tgt", 0), ("hits", 1), ("beg", 2), ("end", 3)):
        assert np.array_equal(merged[:, :, j], cw[f]), f


@pytest.mark.parametrize("quad", ["0", "1"])
@pytest.mark.parametrize("lowest,K", [(0, 2), (0, 4), (4, 2), (4, 3)])
def test_strain_rich_database_mid_lists_against_oracle(tmp_path, lowest, K, quad, monkeypatch):
    """5 species x 8 strains (0.5 % divergence): a 150 bp read collects 50..250 locations over up to 8 targets, pairs more --
    the list lengths handled by mid_cands_kernel (4 / 8 / 16 lanes per query, register bitonic sort), sequence level and
    merged at species level, against the oracle."""
    from metacache_amd import synth
    monkeypatch.setenv("MC_QUAD_LOOKUP", quad)     # both bucket fetch schemes of probe_cands (the second one is for large tables)
    monkeypatch.setenv("MC_COMPACT_LOCATIONS", quad)   # ... and both location stores: 8 bytes per location / 4 (mc_load_location_range)
    rng = np.random.default_rng(4242 + lowest + K)
    genomes, parents = [], []
    for sp in range(5):
        base = synth.random_genome(rng, 60_000)
        for st in range(8):
            genomes.append(synth.mutate(rng, base, 0.005) if st else base)
            parents.append(1000 + sp)
    bld = api.Builder(target_id_bytes=4, max_candidates=K)
    for i, g in enumerate(genomes):
        bld.add_target(g, f"S{i:04d}.1", parent_taxid=parents[i])
    name = str(tmp_path / "strains")
    bld.finish(load=False)
    bld.write(name, [(1, 1, 20, "root")] + [(1000 + i, 1, 4, f"sp{i}") for i in range(5)])
    bld.free()
    reads, _, _ = synth.sample_reads(rng, genomes, 6000, 150, 0.01, 0.002)
    reads = [bytes(r) for r in reads]
    mates = [bytes(synth.revcomp(np.frombuffer(r, dtype=np.uint8)))[:120] for r in reads[:1500]]
    odb = cpuref.oracle().open(name)
    db = api.Database.open(name, max_candidates=K, slot_max_queries=1 << 12, slot_max_chars=1 << 21)
    assert db.table_layout()["location_bytes"] == (4 if quad == "1" else 8)
    cands, counts, _ = db.query(reads, lowest=lowest)
    pc, pcounts, _ = db.query(reads[:1500], mates, lowest=lowest, insert_max=400)
    # window ranges wider than 8 (insert size 1200 => 12 windows): the counting kernel refuses, lists up to 256 are sorted in registers
    pw, _, _ = db.query(reads[:1500], mates, lowest=lowest, insert_max=1200)
    db.close()
    assert 0.5 < np.mean((counts > 32) & (counts <= 256))          # most reads take the mid / counting path
    assert np.any((counts > 32) & (counts <= 64)) and np.any((counts > 64) & (counts <= 128)) and np.any(counts > 128)

    def check(got, i, a, b, ins):
        _, e = odb.query(a, b, K, lowest, ins)
        e = e[:K]
        for j in range(K):
            if j < len(e):
                assert (got[i, j]["tgt"], got[i, j]["hits"], got[i, j]["beg"], got[i, j]["end"]) == \
                       (e[j]["tgt"], e[j]["hits"], e[j]["beg"], e[j]["end"]), (i, j, got[i], e)
            else:
                assert got[i, j]["hits"] == 0, (i, j, got[i], e)
    for i in range(len(reads)):
        check(cands, i, reads[i], b"", 0)
    for i in range(1500):
        check(pc, i, reads[i], mates[i], 400)
        check(pw, i, reads[i], mates[i], 1200)
    odb.close()


@pytest.mark.parametrize("lowest,K", [(0, 1), (0, 2), (0, 4), (4, 2), (4, 3), (6, 4)])
def test_long_lists_hash_cands_against_oracle(tmp_path, lowest, K, monkeypatch):
    """6 species x 20 strains (0.5 % divergence, 2 genera): a 150 bp read collects 150..600 locations, a pair up to ~900 -- the
    lists hash_cands_kernel takes (one wave per query, (target, window) counts in an LDS hash table instead of a sort).  Near-identical
    strains make ties in hits the rule, so the order among equals (ascending target, earliest window range) is what is tested;
    sequence level, merged at species and at genus level, against the oracle."""
    from metacache_amd import synth
    monkeypatch.setenv("MC_COMPACT_LOCATIONS", str(K & 1))          # both location stores (8 / 4 bytes per location)
    monkeypatch.setenv("MC_BIG_MIN", "4096")                        # keep these lists with hash_cands_kernel (by default lists above 128 are filtered first)
    rng = np.random.default_rng(991 + 10 * lowest + K)
    genomes, parents = [], []
    for sp in range(6):
        base = synth.random_genome(rng, 30_000)
        if sp % 2:                                   # a repeat inside the genome: several window ranges of one target compete
            base[20_000:20_560] = base[2_240:2_800]
        for st in range(20):
            genomes.append(synth.mutate(rng, base, 0.005) if st else base)
            parents.append(1000 + sp)
    bld = api.Builder(target_id_bytes=4, max_candidates=K)
    for i, g in enumerate(genomes):
        bld.add_target(g, f"S{i:04d}.1", parent_taxid=parents[i])
    name = str(tmp_path / "manystrains")
    bld.finish(load=False)
    taxa = [(1, 1, 20, "root"), (500, 1, 6, "genus a"), (501, 1, 6, "genus b")] + [(1000 + i, 500 + i % 2, 4, f"sp{i}") for i in range(6)]
    bld.write(name, taxa)
    bld.free()
    reads, _, _ = synth.sample_reads(rng, genomes, 4000, 150, 0.01, 0.002)
    reads = [bytes(r) for r in reads]
    mates = [bytes(synth.revcomp(np.frombuffer(r, dtype=np.uint8)))[:120] for r in reads[:1000]]
    odb = cpuref.oracle().open(name)
    db = api.Database.open(name, max_candidates=K, slot_max_queries=1 << 12, slot_max_chars=1 << 21)
    assert db.table_layout()["location_bytes"] == (4 if K & 1 else 8)
    cands, counts, _ = db.query(reads, lowest=lowest)
    pc, pcounts, _ = db.query(reads[:1000], mates, lowest=lowest, insert_max=400)
    db.close()
    assert 0.3 < np.mean((counts > 256) & (counts <= 1024)) and 0.5 < np.mean((pcounts > 256) & (pcounts <= 1024))

    def check(got, i, a, b, ins):
        _, e = odb.query(a, b, K, lowest, ins)
        e = e[:K]
        for j in range(K):
            if j < len(e):
                assert (got[i, j]["tgt"], got[i, j]["hits"], got[i, j]["beg"], got[i, j]["end"]) == \
                       (e[j]["tgt"], e[j]["hits"], e[j]["beg"], e[j]["end"]), (i, j, got[i], e)
            else:
                assert got[i, j]["hits"] == 0, (i, j, got[i], e)
    for i in range(len(reads)):
        check(cands, i, reads[i], b"", 0)
    for i in range(1000):
        check(pc, i, reads[i], mates[i], 400)
    odb.close()


@pytest.mark.parametrize("quad", ["0", "1"])
@pytest.mark.parametrize("lowest,K", [(0, 2), (4, 4)])
def test_long_reads_chunk_lanes_against_oracle(golden, lowest, K, quad, monkeypatch):
    """Single reads above 512 bp are cut into chunks of 4 windows, each sketched and probed by its own lane (chunk_sketch_kernel /
    chunk_probe_kernel), then sorted by the wave kernel: 513 .. 12000 bp, with substitutions, N runs and lower case, mixed with
    short reads in one batch, against the oracle (which is pinned to the reference)."""
    from metacache_amd import synth
    monkeypatch.setenv("MC_QUAD_LOOKUP", quad)
    monkeypatch.setenv("MC_COMPACT_LOCATIONS", quad)
    rng = np.random.default_rng(777 + K)
    names = [golden.db_path("toy32")]
    odb = cpuref.oracle().open(names[0])
    # the toy genomes are not stored in the fixtures: long reads = concatenations of the golden reads (real k-mers of the targets)
    single, _, _ = golden.reads()
    pool = [r for r in single[:1500] if len(r) == 150]
    reads = []
    for L in [513, 514, 560, 575, 576, 577, 1024, 1025, 2000, 4095, 4096, 4097, 9000, 12000] + [int(x) for x in rng.integers(520, 6000, 60)]:
        parts, tot = [], 0
        while tot < L:
            p = pool[int(rng.integers(0, len(pool)))]
            parts.append(p); tot += len(p)
        r = bytearray(b"".join(parts)[:L])
        for _ in range(int(L * 0.03)):
            r[int(rng.integers(0, L))] = b"ACGT"[int(rng.integers(0, 4))]
        if rng.random() < 0.3:
            a = int(rng.integers(0, L - 40)); r[a:a + 30] = b"N" * 30
        if rng.random() < 0.2:
            r = bytearray(bytes(r).lower())
        reads.append(bytes(r))
    reads += [bytes(p) for p in pool[:200]]                                # short reads in the same batches
    db = api.Database.open(names[0], max_candidates=K, slot_max_queries=64, slot_max_chars=1 << 18)
    cands, counts, _ = db.query(reads, lowest=lowest)
    db.close()
    for i, r in enumerate(reads):
        h, e = odb.query(r, b"", K, lowest, 0)
        assert counts[i] == len(h), (i, len(r))
        e = e[:K]
        for j in range(K):
            if j < len(e):
                assert (cands[i, j]["tgt"], cands[i, j]["hits"], cands[i, j]["beg"], cands[i, j]["end"]) == \
                       (e[j]["tgt"], e[j]["hits"], e[j]["beg"], e[j]["end"]), (i, len(r), j, cands[i], e)
            else:
                assert cands[i, j]["hits"] == 0, (i, j)
    odb.close()


def test_size_independent_properties_at_bench_scale(tmp_path):
    """10^6 reads of the bench workload shape (random genomes, 150 bp, 1 % substitutions): properties that need no oracle --
    (1) an all-N batch and a batch of reads shorter than k produce no candidates at all;
    (2) batching is invisible: the same reads in batches of 65 536 and of 1 000 000 give identical results;
    (3) the lane / mid kernels and the wave kernels agree (MC_NO_LANE_PATH=1);
    (4) every sorted location list is non-decreasing and as long as the reported hit count, candidates are ordered by hits."""
    import os
    import torch
    from metacache_amd import synth
    import bench
    dev = torch.device("cuda", 0)
    G, GL, n = 8, 1_000_000, 1_000_000
    genomes = bench.make_genomes(G, GL, seed=5)
    bld = api.Builder(target_id_bytes=2, max_candidates=2)
    for i, g in enumerate(genomes):
        bld.add_target(g, f"P{i}", parent_taxid=1000 + i)
    bld.finish(load=False)
    name = str(tmp_path / "prop")
    bld.write(name, [(1, 1, 20, "root")] + [(1000 + i, 1, 4, f"sp{i}") for i in range(G)])
    bld.free()
    gcat = torch.from_numpy(np.concatenate(genomes)).to(dev)
    goff = torch.arange(G, device=dev, dtype=torch.int64) * GL
    reads = bench.synth_reads_gpu(gcat, goff, GL, n, seed=77)                       # [n, 152] ASCII, zero padded
    allN = torch.zeros_like(reads)
    allN[:, :150] = ord("N")
    qinfo = torch.zeros((n, 4), dtype=torch.int32, device=dev)
    qinfo[:, 0] = torch.arange(n, device=dev, dtype=torch.int32) * 152
    qinfo[:, 1] = 150
    qinfo[:, 2] = qinfo[:, 0]
    slack = torch.zeros(16, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()

    def run(db, batch, lo, hi, allhits=False):
        m = hi - lo
        seq = torch.cat([batch[lo:hi].reshape(-1), slack])
        qi = qinfo[:m].contiguous()
        out = torch.zeros((m, 2, 4), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        r = db.query_device(seq.data_ptr(), qi.data_ptr(), m, m * 152, max_win_uniform=3, want_allhits=allhits)
        db.copy_results(out.data_ptr(), r.cands, m * 32)
        extra = None
        if allhits:
            off = torch.zeros(m + 1, dtype=torch.int64, device=dev)
            db.copy_results(off.data_ptr(), r.hit_offsets, (m + 1) * 8)
            db.synchronize()
            hits = torch.zeros(int(off[-1]), dtype=torch.int64, device=dev)
            db.copy_results(hits.data_ptr(), r.hits, hits.numel() * 8)
            extra = (off, hits)
        db.synchronize()
        return out, extra

    db = api.Database.open(name, max_candidates=2)
    whole, _ = run(db, reads, 0, n)
    assert int((whole[:, 0, 1] > 0).sum()) > 0.9 * n
    empty, _ = run(db, allN, 0, 100_000)
    assert int(empty[:, :, 1].abs().sum()) == 0                                     # (1)
    qinfo[:, 1] = 15                                                                # shorter than k = 16
    torch.cuda.synchronize()
    short, _ = run(db, reads, 0, 100_000)
    assert int(short[:, :, 1].abs().sum()) == 0
    qinfo[:, 1] = 150
    torch.cuda.synchronize()
    for lo in range(0, n, 65536 * 4):                                               # (2), every fourth small batch
        part, _ = run(db, reads, lo, min(n, lo + 65536))
        assert torch.equal(part, whole[lo:min(n, lo + 65536)])
    assert bool((whole[:, 0, 1] >= whole[:, 1, 1]).all())                           # (4) candidates by hits
    sub, (off, hits) = run(db, reads, 0, 200_000, allhits=True)                     # wave kernels + sorted lists
    assert torch.equal(sub, whole[:200_000])                                        # (3) -allhits takes the wave path
    cnt = off[1:] - off[:-1]
    nondecr = hits[1:] >= hits[:-1]
    starts = off[1:-1]                                                              # first element of every list but the first
    nondecr[(starts - 1)[(starts > 0) & (starts < hits.numel())]] = True            # list boundaries do not count
    assert bool(nondecr.all())
    assert bool((sub[:, 0, 1].long() <= cnt).all())
    db.close()
    os.environ["MC_NO_LANE_PATH"] = "1"
    try:
        dbw = api.Database.open(name, max_candidates=2)
        wave, _ = run(dbw, reads, 0, 300_000)
        dbw.close()
    finally:
        os.environ.pop("MC_NO_LANE_PATH", None)
    assert torch.equal(wave, whole[:300_000])                                       # (3)
    os.environ["MC_QUAD_LOOKUP"] = "1"                                              # the bucket fetch scheme for large tables
    try:
        dbq = api.Database.open(name, max_candidates=2)
        quad, _ = run(dbq, reads, 0, n)
        dbq.close()
    finally:
        os.environ.pop("MC_QUAD_LOOKUP", None)
    assert torch.equal(quad, whole)


def test_location_range_is_enforced_and_optional():
    """mc_load_target_windows / mc_load_location_range: a location outside its target's announced windows fails the load loudly (the
    compact store numbers windows globally and could not hold it) -- single locations, which stay in their buckets, included; without
    the call, and with more windows than 32 bits can number, the table keeps 8-byte locations."""
    import ctypes as C
    L = api.lib()
    keys = np.array([11, 22, 33], dtype=np.uint32)
    sizes = np.array([2, 1, 3], dtype=np.uint8)
    vals = np.array([[5, 0], [9, 1], [7, 2], [1, 0], [2, 1], [300, 2]], dtype=np.uint32)      # {win, tgt}; the last window is 300

    def load(rng=None, windows=None):
        cfg = api.default_config(target_id_bytes=4)
        h = C.c_void_p()
        assert L.mc_create(C.byref(cfg), C.byref(h)) == 0
        if rng:
            assert L.mc_load_location_range(h, rng[0], rng[1]) == 0
        if windows is not None:
            w = np.asarray(windows, dtype=np.uint32)
            assert L.mc_load_target_windows(h, w.ctypes.data, len(w)) == 0
        assert L.mc_load_begin(h, 0, 3, 6) == 0
        rc = L.mc_load_batch(h, 0, keys.ctypes.data, sizes.ctypes.data, vals.ctypes.data, 3)
        rc = rc or L.mc_load_end(h, 0)
        lay = (C.c_uint64 * 4)()
        L.mc_table_layout(h, lay)
        err = L.mc_last_error(h).decode()
        L.mc_destroy(h)
        return rc, int(lay[0]), int(lay[1]) & 0xFFFFFFFF, err              # (layout[1]: low 32 bits = the gap, high = the list alignment)
    assert load()[:2] == (0, 8)
    assert load((2, 300))[:3] == (0, 4, 1024)                     # (layout[1] = the gap between two targets' window numbers)
    assert load((2, 511))[:2] == (0, 4)
    rc, _, _, err = load((2, 299))
    assert rc != 0 and "range" in err
    assert load(windows=[6, 10, 301])[:2] == (0, 4)               # every target exactly as long as its last window
    rc, _, _, err = load(windows=[6, 10, 300])
    assert rc != 0 and "range" in err                             # window 300 of target 2
    rc, _, _, err = load(windows=[6, 9, 301])
    assert rc != 0 and "range" in err                             # the SINGLE location (window 9 of target 1): range-checked as well
    rc, _, _, err = load(windows=[6, 10])
    assert rc != 0 and "range" in err                             # unknown target
    assert load((1 << 20, 1 << 20))[:2] == (0, 8)                  # 2^40 windows: stays wide
    assert load((0xFFFF, 0xFFFE))[:2] == (0, 8)                    # 2^32 windows + gaps: stays wide
    assert load((0xFFFF, 0x7FFF))[:2] == (0, 4)


@pytest.mark.parametrize("lowest,K", [(0, 2), (0, 3), (4, 2)])
def test_two_regions_of_one_target_in_the_filtered_path(tmp_path, lowest, K, monkeypatch):
    """A 400-bp segment that occurs TWICE in every strain's genome, 200 kbp (1 780 windows) apart: reads of it find two regions per
    target with equal hits.  gw_count_kernel strikes a winner's REGION (numbers within 1 024 of it) and looks the winners' targets up
    afterwards; here its second winner is the first winner's other region, which it must notice and hand the read to the exact path
    (the sorted path's kernel sees whole target runs).  12 strains x 2 regions x 32 features: lists of several hundred locations."""
    from metacache_amd import synth
    monkeypatch.setenv("MC_COMPACT_LOCATIONS", "1")
    monkeypatch.setenv("MC_BIG_MIN", "0")
    rng = np.random.default_rng(99 + lowest + K)
    genomes, parents = [], []
    for sp in range(2):
        base = synth.random_genome(rng, 260_000)
        base[201_000:201_400] = base[1_000:1_400]
        for st in range(12):
            genomes.append(synth.mutate(rng, base, 0.004) if st else base.copy())
            parents.append(1000 + sp)
    bld = api.Builder(target_id_bytes=4, max_candidates=K)
    for i, g in enumerate(genomes):
        bld.add_target(g, f"R{i:04d}.1", parent_taxid=parents[i])
    name = str(tmp_path / "regions")
    bld.finish(load=False)
    bld.write(name, [(1, 1, 20, "root")] + [(1000 + i, 1, 4, f"sp{i}") for i in range(2)])
    bld.free()
    reads = []
    for _ in range(1500):
        g = genomes[int(rng.integers(0, len(genomes)))]
        st = int(rng.integers(1_000, 1_250)) + (200_000 if rng.random() < 0.5 else 0)
        reads.append(bytes(synth.mutate(rng, g[st:st + 150], 0.01, 0.001)))
    others, _, _ = synth.sample_reads(rng, genomes, 1500, 150, 0.01, 0.002)
    reads += [bytes(r) for r in others]
    mates = [bytes(synth.revcomp(np.frombuffer(r, dtype=np.uint8)))[:130] for r in reads[:600]]
    odb = cpuref.oracle().open(name)
    db = api.Database.open(name, max_candidates=K, slot_max_queries=1 << 12, slot_max_chars=1 << 21)
    assert db.table_layout()["location_bytes"] == 4
    cands, counts, _ = db.query(reads, lowest=lowest)
    pc, _, _ = db.query(reads[:600], mates, lowest=lowest, insert_max=0)
    st = db.last_batch_stats()
    db.close()
    assert np.mean(counts[:1500] > 256) > 0.5, np.percentile(counts[:1500], [5, 50, 95])
    for i, r in enumerate(reads):
        _, e = odb.query(r, b"", K, lowest, 0)
        assert cands_equal(cands[i], e[:K]), (i, counts[i], cands[i], e[:K])
    for i in range(600):
        _, e = odb.query(reads[i], mates[i], K, lowest, 0)
        assert cands_equal(pc[i], e[:K]), ("pair", i, pc[i], e[:K])
    odb.close()
