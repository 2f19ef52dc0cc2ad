"""GPU: the minimal database builder (mc_build_*): bucket contents against an independent construction
from the oracle's sketcher, the written files against the oracle's (and, if present, the reference's)
reader, and queries on the directly-loaded table against queries on the files."""
import os

import numpy as np
import pytest

import cpuref
from metacache_amd import api, synth

pytestmark = pytest.mark.gpu


def make_genomes(rng, n=20, length=10000, copies=26):
    repeat = synth.random_genome(rng, 448)
    gs = []
    for _ in range(n):
        g = synth.random_genome(rng, length + int(rng.integers(0, 200)))
        for slot in rng.choice(np.arange(1, length // 112 - 5), size=copies, replace=False):
            p = int(slot) * 112
            g[p:p + 448] = repeat
        gs.append(g)
    gs.append(synth.random_genome(rng, 100))      # shorter than one window
    gs.append(synth.random_genome(rng, 127))      # exactly one window
    gs.append(synth.random_genome(rng, 10))       # shorter than k: no windows at all
    gs.append(np.frombuffer(b"N" * 300, dtype=np.uint8).copy())   # windows without features still count
    gs.append(synth.random_genome(rng, 112 * 300 + 127))          # > 256 windows: crosses a builder chunk boundary
    return gs


@pytest.mark.parametrize("tb", [4, 2])
def test_builder_buckets_files_and_queries(tmp_path, tb):
    rng = np.random.default_rng(5 + tb)
    orc = cpuref.oracle()
    genomes = make_genomes(rng)
    bld = api.Builder(target_id_bytes=tb, max_candidates=2, copy_allhits=1)
    for i, g in enumerate(genomes):
        bld.add_target(g, f"SYN_{i:05d}.1", parent_taxid=1000 + i % 5, filename=f"f{i}.fa")
    db = bld.finish(load=True)
    name = str(tmp_path / "built")
    taxa = [(1, 1, 20, "root")] + [(1000 + i, 1, 4, f"species {i}") for i in range(5)]
    bld.write(name, taxa)
    bld.free()

    # independent expectation: feature -> first 254 (tgt, win) in insertion order
    expect = {}
    for t, g in enumerate(genomes):
        feats, counts = orc.sketch(g.tobytes(), 16, 16, 127, 112)
        for w in range(len(counts)):
            for f in feats[w, :counts[w]]:
                expect.setdefault(int(f), []).append((t << 32) | w)
    odb = orc.open(name)
    assert odb.ref.lib.mco_db_target_id_bytes(odb.h) == tb
    assert odb.n_targets == len(genomes)
    nloc = 0
    for f, locs in expect.items():
        got = odb.lookup(f)
        want = np.array(locs[:254], dtype=np.uint64)
        assert np.array_equal(got, want), f
        nloc += len(want)
    assert odb.n_locations == nloc == db.n_locations
    assert max(len(v) for v in expect.values()) > 254          # the cap was exercised

    # queries: table loaded straight from the builder == table loaded from the files == oracle on the files
    reads, _, _ = synth.sample_reads(rng, [g for g in genomes if g.size > 200], 600, 150, 0.01, 0.002)
    reads = [bytes(r) for r in reads]
    c1, n1, h1 = db.query(reads)
    db2 = api.Database.open(name, max_candidates=2, copy_allhits=1)
    c2, n2, h2 = db2.query(reads)
    assert np.array_equal(c1, c2) and np.array_equal(n1, n2)
    have_ref = cpuref.have_reference(tb)
    rdb = cpuref.reference(tb).open(name) if have_ref else None
    for i, r in enumerate(reads):
        h, c = odb.query(r, b"", 2, 0, 0)
        assert np.array_equal(h1[i]["win"], h["win"]) and np.array_equal(h1[i]["tgt"], h["tgt"]), i
        assert np.array_equal(h2[i]["win"], h["win"]) and np.array_equal(h2[i]["tgt"], h["tgt"]), i
        k = len(c)
        assert int((c1[i]["hits"] > 0).sum()) == k
        for f in ("tgt", "hits", "beg", "end"):
            assert np.array_equal(c1[i][:k][f], c[f]), (i, f)
        if rdb is not None:                                   # the real reference reads OUR files
            hr, cr = rdb.query(r, b"", 2, 0, 0)
            assert np.array_equal(hr, h) and np.array_equal(cr, c), i
    if rdb is not None:
        assert rdb.info()[:7] == odb.info()[:7]
        assert np.array_equal(rdb.lineages(), odb.lineages())
        rdb.close()
    odb.close(); db.close(); db2.close()


def test_key_sharded_builders_make_the_same_table(tmp_path):
    """mc_build_finish_shards: three builders, each keeping one key shard of the same targets, loaded into ONE table -- every query
    answers as with the table of a single builder (sorted location lists and candidates)."""
    rng = np.random.default_rng(77)
    genomes = make_genomes(rng)
    whole = api.Builder(target_id_bytes=4, max_candidates=3, copy_allhits=1)
    shards = [api.Builder(target_id_bytes=4, max_candidates=3, copy_allhits=1, key_shard_index=i, key_shard_count=3) for i in range(3)]
    for i, g in enumerate(genomes):
        for b in [whole] + shards:
            b.add_target(g, f"SYN_{i:05d}.1", parent_taxid=1000 + i % 5, filename=f"f{i}.fa")
    db1 = whole.finish(load=True)
    db3 = api.Builder.finish_shards(shards)
    assert db1.info()[7] == db3.info()[7] and db1.info()[5] == db3.info()[5]          # locations, targets
    reads, _, _ = synth.sample_reads(rng, [g for g in genomes if g.size > 2000], 1500, 150, 0.01, 0.002)
    reads = [bytes(r) for r in reads]
    c1, n1, h1 = db1.query(reads, lowest=0)
    c3, n3, h3 = db3.query(reads, lowest=0)
    assert np.array_equal(n1, n3)
    for i in range(len(reads)):
        assert np.array_equal(h1[i], h3[i]), i
        assert np.array_equal(c1[i], c3[i]), i
    with pytest.raises(api.McError):
        shards[0].write(str(tmp_path / "partial"), [(1, 1, 20, "root")])              # a shard is not a database
    db1.close(); db3.close()
    for b in [whole] + shards:
        b.free()


@pytest.mark.parametrize("shards,max_ambig", [(1, 1), (1, 2), (3, 1), (1, 0)])
def test_remove_ambiguous_features(tmp_path, shards, max_ambig):
    """mc_build_remove_ambiguous (host_hashmap.hpp:499-540): a feature stays iff its (capped) location list touches at most max_ambig
    different taxa; targets without a taxon on the rank count as one taxon; 0 means 1.  Checked on the written files."""
    rng = np.random.default_rng(300 + shards + max_ambig)
    orc = cpuref.oracle()
    genomes = make_genomes(rng)
    anc = np.array([0 if t % 7 == 3 else 1 + t % 4 for t in range(len(genomes))], dtype=np.uint32)
    bs = [api.Builder(target_id_bytes=4, key_shard_index=i, key_shard_count=shards) for i in range(shards)]
    for i, g in enumerate(genomes):
        for b in bs:
            b.add_target(g, f"SYN_{i:05d}.1", parent_taxid=0, filename=f"f{i}.fa")
    before = after = removed = 0
    for b in bs:
        b.finish(load=False)
        before += b.counts()[0]
        removed += b.remove_ambiguous(anc, max_ambig)
        after += b.counts()[0]
    expect = {}
    for t, g in enumerate(genomes):
        feats, counts = orc.sketch(g.tobytes(), 16, 16, 127, 112)
        for w in range(len(counts)):
            for f in feats[w, :counts[w]]:
                expect.setdefault(int(f), []).append((t << 32) | w)
    lim = max(max_ambig, 1)
    kept = {f: l[:254] for f, l in expect.items() if len({int(anc[v >> 32]) for v in l[:254]}) <= lim}
    assert before == len(expect) and after == len(kept) and removed == before - after and 0 < after < before
    name = str(tmp_path / "ambig")
    from metacache_amd.api import lib
    import ctypes as C
    arr = (C.c_void_p * shards)(*[b.h for b in bs])
    lib().mc_build_write_shards.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.c_char_p, C.c_void_p, C.c_uint64]
    assert lib().mc_build_write_shards(arr, shards, name.encode(), None, 0) == 0
    odb = orc.open(name)
    assert odb.n_locations == sum(len(l) for l in kept.values())
    for f, l in expect.items():
        got = odb.lookup(f)
        assert np.array_equal(got, np.array(kept.get(f, []), dtype=np.uint64)), f
    # the table loaded from the builders answers like the files
    db = api.Builder.finish_shards(bs)
    assert db.n_locations == odb.n_locations
    odb.close(); db.close()
    for b in bs:
        b.free()


def test_streaming_build_reuses_and_releases_large_blocks():
    """devcache.cpp: between mc_build_table_begin and _end the key shards' large scratch buffers (64 MB and more) are kept for the next
    shard instead of going back through hipMalloc; afterwards nothing is kept -- the device's free memory after closing the database is
    what it was before the build -- and the table answers like the one of a single-pass build."""
    import torch
    from metacache_amd import synthdb
    spec = synthdb.phylogeny(12, 2, 2, 4_000_000, 5_000_000, seed=77)          # 48 targets, ~0.2 Gbp: shard buffers of several hundred MB
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info(0)
    db4, info4 = synthdb.build_database(spec, shards=4, max_candidates=2)
    db1, info1 = synthdb.build_database(spec, shards=1, max_candidates=2)
    P = synthdb.read_params(spec, 78)
    reads = [bytes(r[:150]) for r in synthdb.CpuSynth().reads(spec, P, 0, 3000)]
    c4, n4, _ = db4.query(reads)
    c1, n1, _ = db1.query(reads)
    assert np.array_equal(n4, n1)
    for f in ("tgt", "hits", "beg", "end"):
        assert np.array_equal(c4[f], c1[f]), f
    db4.close(); db1.close()
    torch.cuda.empty_cache(); torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info(0)
    assert free0 - free1 < (512 << 20), (free0, free1)              # (nothing of the builds is still held: 64 GB could be, were the cache left open)
