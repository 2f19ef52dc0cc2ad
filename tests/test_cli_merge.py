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
