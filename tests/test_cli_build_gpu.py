"""`mcq build` / `mcq build+query` (metacache_amd/csrc/mcq_build.h) against what the reference's own `build` made of the same
sequence files, taxonomy dumps and id tables (tests/golden/build_expected.json.gz, made by tests/golden/make_golden_build.py from
oracle/_ref/metacache_u32 / _u16): the database FILES are compared record by record (taxa, sources, window counts, every feature's
location list; the order inside the files is free -- the reference writes its hash tables in iteration order), the query output on
the new database line by line."""
import gzip
import json
import os
import struct
import subprocess

import numpy as np
import pytest

from metacache_amd import build
from test_cli_gpu import _same

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")

with gzip.open(os.path.join(GOLD, "build_expected.json.gz"), "rt") as f:
    EXP = json.load(f)


def parse_db(name):
    """order-free content of <name>.meta / .cache0 (database.cpp:247-290, hash_multimap.hpp:1037-1082)"""
    b = open(name + ".meta", "rb").read()
    p = 0

    def rd(fmt):
        nonlocal p
        v = struct.unpack_from("<" + fmt, b, p)
        p += struct.calcsize("<" + fmt)
        return v if len(v) > 1 else v[0]

    def rstr():
        nonlocal p
        n = rd("Q")
        s = b[p:p + n].decode()
        p += n
        return s
    ver = rd("Q")
    widths = rd("7B")
    sk1 = rd("4Q"); sk2 = rd("4Q")
    maxlocs = rd("Q")
    ntgt = rd("H") if widths[1] == 2 else rd("I")
    nparts = rd("I")
    ntaxa = rd("Q")
    taxa = {}
    for _ in range(ntaxa):
        tid, parent, rank = rd("q"), rd("q"), rd("B")
        nm, fn = rstr(), rstr()
        idx, win = rd("Q"), rd("Q")
        assert str(tid) not in taxa
        taxa[str(tid)] = [parent, rank, nm, fn, idx, win]
    assert p == len(b)
    c = open(name + ".cache0", "rb").read()
    nkeys, nvals, batch = struct.unpack_from("<3Q", c, 0)
    p = 24
    tb = widths[1]
    feats = {}
    done = 0
    while done < nkeys:
        nb = min(batch, nkeys - done)
        keys = np.frombuffer(c, dtype="<u4", count=nb, offset=p); p += 4 * nb
        sizes = np.frombuffer(c, dtype=np.uint8, count=nb, offset=p); p += nb
        for k, s in zip(keys.tolist(), sizes.tolist()):
            vals = []
            for _ in range(s):
                win = struct.unpack_from("<I", c, p)[0]
                tgt = struct.unpack_from("<H" if tb == 2 else "<I", c, p + 4)[0]
                p += 4 + tb
                vals.append([tgt, win])
            assert str(k) not in feats and s > 0
            feats[str(k)] = vals
        done += nb
    assert p == len(c) and sum(len(v) for v in feats.values()) == nvals
    return {"version": ver, "widths": list(widths), "sketching": list(sk1), "sketching2": list(sk2), "maxlocs": maxlocs, "targets": ntgt,
            "parts": nparts, "taxa": taxa, "features": feats}


def same_db(got, exp, tag):
    for k in ("version", "widths", "sketching", "sketching2", "maxlocs", "targets", "parts"):
        assert got[k] == exp[k], (tag, k, got[k], exp[k])
    assert set(got["taxa"]) == set(exp["taxa"]), (tag, sorted(set(got["taxa"]) ^ set(exp["taxa"]))[:10])
    for k, v in exp["taxa"].items():
        assert got["taxa"][k] == v, (tag, k, got["taxa"][k], v)
    assert set(got["features"]) == set(exp["features"]), (tag, len(got["features"]), len(exp["features"]))
    for k, v in exp["features"].items():
        assert got["features"][k] == v, (tag, k)


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(EXP["build"]))
def test_build_writes_the_reference_database(case, tmp_path):
    build.build_library()
    c = EXP["build"][case]
    db = str(tmp_path / case)
    r = subprocess.run([build.MCQ, "build", db] + c["args"] + c["mcq_extra"], cwd=GOLD, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    same_db(parse_db(db), c["db"], case)
    # ... and queried through the files it wrote: every line of the reference's output on ITS database
    out = tmp_path / "q.txt"
    r = subprocess.run([build.MCQ, "query", db, "build_reads.fa"] + EXP["query_args"] + ["-threads", "1", "-out", str(out)], cwd=GOLD,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    _same(out.read_text().split("\n"), c["query"], case)


@pytest.mark.gpu
@pytest.mark.parametrize("shards", [1, 3])
@pytest.mark.parametrize("case", sorted(EXP.get("modify", {})))
def test_modify_adds_to_a_database(case, shards, tmp_path):
    """mcq modify: the database mcq built from the first files is read back into a builder (targets, location lists), the other files
    are added, taxonomy / ranking / feature removal run over everything -- the files equal those the reference's build + modify wrote"""
    build.build_library()
    c = EXP["modify"][case]
    db = str(tmp_path / case)
    more = ["-build-shards", str(shards)] if shards > 1 else []
    for mode, args in (("build", c["first"]), ("modify", c["second"])):
        r = subprocess.run([build.MCQ, mode, db] + args + (c["mcq_extra"] if mode == "build" else []) + more, cwd=GOLD, capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0, (mode, r.stderr)
    same_db(parse_db(db), c["db"], case)
    out = tmp_path / "q.txt"
    r = subprocess.run([build.MCQ, "query", db, "build_reads.fa"] + EXP["query_args"] + ["-threads", "1", "-out", str(out)], cwd=GOLD,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    _same(out.read_text().split("\n"), c["query"], case)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["default", "overpopulated", "u16", "ambig_species"])
def test_build_in_key_shards_writes_the_same_database(case, tmp_path):
    """-build-shards 3: three builders keep a third of the features each (what inputs beyond 2^32 (feature, location) pairs get
    automatically); mc_build_write_shards puts them into one file set with the content of the reference's database"""
    build.build_library()
    c = EXP["build"][case]
    db = str(tmp_path / case)
    r = subprocess.run([build.MCQ, "build", db] + c["args"] + c["mcq_extra"] + ["-build-shards", "3"], cwd=GOLD, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    assert "3 key shards" in r.stdout
    same_db(parse_db(db), c["db"], case)


@pytest.mark.gpu
def test_build_query_in_key_shards(tmp_path):
    build.build_library()
    c = EXP["bq"]["bq_species"]
    out = tmp_path / "bq.txt"
    r = subprocess.run([build.MCQ, "build+query"] + c["args"] + ["-build-shards", "2", "-threads", "1", "-out", str(out)],
                       cwd=GOLD, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    _same(out.read_text().split("\n"), c["lines"], "bq_species/shards")


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(EXP["bq"]))
def test_build_query_in_memory(case, tmp_path):
    """build+query: the table is filled from the builder's device arrays, no database file in between (-save-db writes one afterwards)"""
    build.build_library()
    c = EXP["bq"][case]
    out = tmp_path / "bq.txt"
    saved = str(tmp_path / "saved")
    r = subprocess.run([build.MCQ, "build+query"] + [a.format(savedb=saved) for a in c["args"]] + ["-threads", "1", "-out", str(out)],
                       cwd=GOLD, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    _same(out.read_text().split("\n"), c["lines"], case)
    if "saved" in c:
        same_db(parse_db(saved), c["saved"], case)
    else:
        assert not os.path.exists(saved + ".meta")


@pytest.mark.gpu
def test_build_query_interactive(tmp_path):
    """build+query without -query: the interactive loop over the freshly built table; candidate limits change between lines (the
    query context is re-created from the builder's arrays)"""
    build.build_library()
    c = EXP["bq"]["bq_default"]
    bargs = c["args"][:c["args"].index("-query")]
    o1, o2 = tmp_path / "i1.txt", tmp_path / "i2.txt"
    stdin = f"build_reads.fa -tophits -queryids -taxids -out {o1}\nbuild_reads.fa -tophits -queryids -taxids -maxcand 3 -allhits -out {o2}\n\n"
    r = subprocess.run([build.MCQ, "build+query"] + bargs + ["-threads", "1"], cwd=GOLD, input=stdin, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    assert "Running in interactive mode" in r.stdout
    _same(o1.read_text().split("\n"), c["lines"], "interactive-1")
    assert o2.exists() and len(o2.read_text().split("\n")) == len(c["lines"])


def test_build_fails_loudly(tmp_path):
    build.build_library()
    r = subprocess.run([build.MCQ, "build", str(tmp_path / "x")], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "ABORT" in r.stderr
    r = subprocess.run([build.MCQ, "build", str(tmp_path / "x"), os.path.join(GOLD, "build_reads.fa"), "-parts", "2"], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "ABORT" in r.stderr
    import torch
    if not torch.cuda.is_available():                                           # no GPU: no files, no fallback
        r = subprocess.run([build.MCQ, "build", str(tmp_path / "x"), os.path.join(GOLD, "build_reads.fa")], capture_output=True, text=True, timeout=60)
        assert r.returncode != 0 and "ABORT" in r.stderr and not os.path.exists(str(tmp_path / "x.meta"))
