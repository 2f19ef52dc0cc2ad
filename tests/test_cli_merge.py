"""`mcq merge` (mode_merge.cpp: result files of queries against single database parts -> one classification) against the output of
the reference's own `metacache merge` on the same per-part result files (tests/golden/merge_in, made by the reference; expected
lines in tests/golden/merge_expected.json.gz, generator tests/golden/make_golden_merge.py).  Pure host work: runs without a GPU."""
import gzip
import json
import os
import subprocess

import pytest

from metacache_amd import build
from test_cli_gpu import _same

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")

with gzip.open(os.path.join(GOLD, "merge_expected.json.gz"), "rt") as f:
    EXP = json.load(f)


@pytest.mark.parametrize("case", sorted(EXP))
def test_merge_matches_reference(case, tmp_path):
    build.build_library()
    c = EXP[case]
    files = c["files"]
    if case == "directory":
        d = tmp_path / "merge_dir"
        d.mkdir()
        for part in (0, 1):
            os.symlink(os.path.join(GOLD, f"merge_in/part{part}.txt"), d / f"part{part}.txt")
        files = [str(d)]
    out = tmp_path / "merged.txt"
    r = subprocess.run([build.MCQ, "merge"] + files + ["-taxonomy", "build_in/taxonomy"] + c["args"] + ["-out", str(out)], cwd=GOLD,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    got = out.read_text().split("\n")
    exp = c["lines"]
    if case == "directory":                       # the file names are part of the output
        got = [l for l in got if "part" not in l or not l.startswith("# ")]
        exp = [l for l in exp if "part" not in l or not l.startswith("# ")]
    _same(got, exp, case)


def test_merge_refuses_sequence_level_and_single_files(tmp_path):
    build.build_library()
    r = subprocess.run([build.MCQ, "merge", "merge_in/part0.txt", "-taxonomy", "build_in/taxonomy"], cwd=GOLD, capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "ABORT" in r.stderr
    seq = tmp_path / "seq.txt"
    seq.write_text("# Classification will be constrained to ranks from 'sequence' to 'domain'.\n# TABLE_LAYOUT: query_id\t|\tquery_header\t|\ttop_hits\t|\trank:taxname\n")
    r = subprocess.run([build.MCQ, "merge", str(seq), "merge_in/part0.txt", "-taxonomy", "build_in/taxonomy"], cwd=GOLD, capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "sequence level" in r.stderr


# ---- info mode (metadata topics; no GPU needed: mc_open_metadata) -----------------------------------------------------------------
with gzip.open(os.path.join(GOLD, "info_expected.json.gz"), "rt") as f:
    INFO = json.load(f)


@pytest.mark.parametrize("case", sorted(INFO))
def test_info_matches_reference(case):
    """stdout of `mcq info ...` against the reference's: identical except the program version line; the rows of `info rank` are
    compared as a set (the reference walks a map keyed by taxon address)"""
    build.build_library()
    c = INFO[case]
    r = subprocess.run([build.MCQ, "info"] + c["args"], cwd=GOLD, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    drop = lambda ls: [l for l in ls if not l.startswith("MetaCache version")]
    got, exp = drop(r.stdout.split("\n")), drop(c["stdout"])
    if case.startswith("rank"):
        got, exp = sorted(got), sorted(exp)
    assert got == exp, (case, [(g, e) for g, e in zip(got, exp) if g != e][:5], len(got), len(exp))


def test_info_topics_of_the_host_table_are_refused():
    build.build_library()
    r = subprocess.run([build.MCQ, "info", "toy32", "statistics"], cwd=GOLD, capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "ABORT" in r.stderr
