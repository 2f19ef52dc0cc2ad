"""`mcq query` (metacache_amd/csrc/mcq_main.cpp) against the output files the reference's own command line wrote for the
same read files and options (tests/golden/cli_expected.json.gz, made by tests/golden/make_golden_cli.py from
oracle/_ref/metacache_u32).  Every line must be identical except the two wall-clock lines of the summary."""
import gzip
import json
import os
import re
import subprocess

import pytest

from metacache_amd import build

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")


def _cases():
    with gzip.open(os.path.join(GOLD, "cli_expected.json.gz"), "rt") as f:
        return json.load(f)


CASES = _cases()


def _volatile(line: str) -> bool:
    # (times, speeds, and the number of hardware threads of the box the command ran on)
    return re.match(r"^(# |%%)(time:    |speed:   |Using \d+ threads$)", line) is not None


def _same(got, exp, tag, unordered=False):
    assert len(got) == len(exp), (tag, len(got), len(exp))
    if unordered:            # the reference lists reference sequences in the iteration order of an unordered_map
        got = sorted(l for l in got if not _volatile(l))
        exp = sorted(l for l in exp if not _volatile(l))
    for i, (g, e) in enumerate(zip(got, exp)):
        if _volatile(e):
            assert _volatile(g)
            continue
        assert g == e, (tag, i, g[:300], e[:300])


PLAIN = sorted(k for k, v in CASES.items() if "lines" in v and "files" in v)
EXTRA = sorted(k for k, v in CASES.items() if "extra" in v)


@pytest.mark.gpu
@pytest.mark.parametrize("case", EXTRA)
def test_cli_analysis_lists_in_separate_files(case, tmp_path):
    """-hits-per-ref <file> / -abundances <file>: the lists go to their own files, the mappings to -out"""
    build.build_library()
    c = CASES[case]
    extra = {k: str(tmp_path / (case + "." + k)) for k in c["extra"]}
    out = tmp_path / "out.txt"
    cmd = [build.MCQ, "query", "toy32"] + c["files"] + [a.format(**extra) for a in c["args"]] + ["-threads", "1", "-out", str(out)]
    r = subprocess.run(cmd, cwd=GOLD, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    _same(out.read_text().split("\n"), c["main"], (case, "main"))
    for k, exp in c["extra"].items():
        _same(open(extra[k]).read().split("\n"), exp, (case, k), unordered=(k == "targets"))
SPLIT = sorted(k for k, v in CASES.items() if "split" in v)


@pytest.mark.gpu
@pytest.mark.parametrize("case", SPLIT)
def test_cli_split_out_matches_reference(case, tmp_path):
    """-split-out: one output file (parameters, mappings, statistics) per input file / file pair, query ids restart"""
    build.build_library()
    c = CASES[case]
    prefix = str(tmp_path / case)
    cmd = [build.MCQ, "query", "toy32"] + c["files"] + c["args"] + ["-threads", "1", "-split-out", prefix]
    r = subprocess.run(cmd, cwd=GOLD, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    produced = sorted(f for f in os.listdir(tmp_path) if f.startswith(case + "_"))
    assert [f[len(case):] for f in produced] == sorted(c["split"])
    for suffix, exp in c["split"].items():
        _same(open(prefix + suffix).read().split("\n"), exp, (case, suffix))


@pytest.mark.gpu
def test_cli_interactive_mode_matches_reference(tmp_path):
    """no input files on the command line: lines from stdin, the initial options are the defaults, the database stays loaded"""
    build.build_library()
    c = CASES["interactive"]
    stdin = ""
    for i, line in enumerate(c["lines"]):
        stdin += " ".join(line + ["-out", str(tmp_path / f"inter{i}.txt")]) + "\n"
    r = subprocess.run([build.MCQ, "query", "toy32"] + c["initial"] + ["-threads", "1"], cwd=GOLD, input=stdin + "\n", capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    assert "Running in interactive mode" in r.stdout and "Terminate." in r.stdout
    for i, exp in enumerate(c["outputs"]):
        _same(open(tmp_path / f"inter{i}.txt").read().split("\n"), exp, ("interactive", i))


@pytest.mark.gpu
def test_cli_reference_formatting_matrix(tmp_path):
    """The 144 option combinations of the reference's own formatting test (test/run_tests:86-117), typed into ONE interactive
    session: -mapped-only / -separator x -omit-ranks / -queryids x -taxids / -taxids-only x -lineage / -separate-cols."""
    build.build_library()
    c = CASES["format_matrix"]
    stdin = ""
    for i, line in enumerate(c["matrix"]):
        stdin += " ".join(["cli_fmt.fa"] + line + ["-out", str(tmp_path / f"fmt{i}.txt")]) + "\n"
    r = subprocess.run([build.MCQ, "query", "toy32", "-threads", "1"], cwd=GOLD, input=stdin + "\n", capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr
    assert len(c["matrix"]) == 144
    for i, exp in enumerate(c["outputs"]):
        _same(open(tmp_path / f"fmt{i}.txt").read().split("\n"), exp, ("format_matrix", i, c["matrix"][i]))


@pytest.mark.gpu
@pytest.mark.parametrize("case", PLAIN)
def test_cli_matches_reference_output(case, tmp_path):
    build.build_library()
    c = CASES[case]
    out = tmp_path / "out.txt"
    cmd = [build.MCQ, "query", "toy32"] + c["files"] + c["args"] + ["-threads", "1", "-out", str(out)]
    r = subprocess.run(cmd, cwd=GOLD, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    _same(out.read_text().split("\n"), c["lines"], case, unordered=any(a.startswith("-hits-per-") for a in c["args"]))


@pytest.mark.gpu
@pytest.mark.parametrize("cap", ["1", "2", "5"])
@pytest.mark.parametrize("case", ["maxcand_unlimited", "maxcand_unlimited_seq"])
def test_cli_maxcand_unlimited_beyond_the_device_list(case, cap, tmp_path):
    """-maxcand 0 = no limit: a read with more candidates than the device list holds (MCQ_UNLIMITED_CAP shrinks the list so that
    the toy database reaches it) gets them from the host's rows 9-10 over its sorted location list -- same output as the reference."""
    build.build_library()
    c = CASES[case]
    out = tmp_path / "out.txt"
    cmd = [build.MCQ, "query", "toy32"] + c["files"] + c["args"] + ["-threads", "1", "-out", str(out)]
    r = subprocess.run(cmd, cwd=GOLD, capture_output=True, text=True, timeout=600, env=dict(os.environ, MCQ_UNLIMITED_CAP=cap))
    assert r.returncode == 0, r.stderr
    _same(out.read_text().split("\n"), c["lines"], (case, cap))


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["everything_species", "pairseq", "two_files", "fastq_irregular", "fastq_irregular_pairseq", "fasta_plus_lines",
                                  "irregular_with_regular", "cov_percentile_pct_hits_per_ref", "hits_per_ref"])
def test_cli_streaming_index_same_output(case, tmp_path):
    """Plain input files are indexed chunk by chunk while the first batches already run (SeqFile::stream_*); MCQ_STREAM_MIN / _CHUNK
    shrink the thresholds so that the toy files take that path in ~20 chunks, with tiny batches and 4 worker threads."""
    build.build_library()
    c = CASES[case]
    out = tmp_path / "out.txt"
    cmd = [build.MCQ, "query", "toy32"] + c["files"] + c["args"] + ["-threads", "4", "-out", str(out)]
    if "-batch-size" not in c["args"]:
        cmd += ["-batch-size", "23"]
    r = subprocess.run(cmd, cwd=GOLD, capture_output=True, text=True, timeout=600, env=dict(os.environ, MCQ_STREAM_MIN="0", MCQ_STREAM_CHUNK="3000"))
    assert r.returncode == 0, r.stderr
    unordered = any(a.startswith("-hits-per-") for a in c["args"])
    got = [l for l in out.read_text().split("\n") if not _volatile(l) and "threads" not in l]
    exp = [l for l in c["lines"] if not _volatile(l) and "threads" not in l]
    assert (sorted(got) == sorted(exp)) if unordered else (got == exp)


@pytest.mark.gpu
def test_cli_replicate_over_gpus(tmp_path):
    """-replicate n (options.cpp:1155-1163): the table on the GPUs 0 .. n-1, the worker threads dealt out over them; the output does
    not depend on it.  More copies than GPUs must fail loudly, naming the GPU that is missing."""
    import torch
    build.build_library()
    c = CASES["everything_species"]
    ngpu = torch.cuda.device_count()
    for n in (1, 2):
        out = tmp_path / f"out{n}.txt"
        cmd = [build.MCQ, "query", "toy32"] + c["files"] + c["args"] + ["-threads", "4", "-replicate", str(n), "-out", str(out)]
        r = subprocess.run(cmd, cwd=GOLD, capture_output=True, text=True, timeout=600)
        if n <= ngpu:
            assert r.returncode == 0, r.stderr
            got = [l for l in out.read_text().split("\n") if not _volatile(l) and "threads" not in l]
            exp = [l for l in c["lines"] if not _volatile(l) and "threads" not in l]
            assert got == exp
        else:
            assert r.returncode != 0 and "GPU 1" in (r.stdout + r.stderr), (r.stdout[-500:], r.stderr[-500:])


@pytest.mark.gpu
def test_cli_small_batches_same_output(tmp_path):
    """-batch-size only changes how reads are grouped into device batches."""
    build.build_library()
    c = CASES["everything_species"]
    out = tmp_path / "out.txt"
    cmd = [build.MCQ, "query", "toy32"] + c["files"] + c["args"] + ["-threads", "1", "-batch-size", "37", "-out", str(out)]
    r = subprocess.run(cmd, cwd=GOLD, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    got = [l for l in out.read_text().split("\n") if not _volatile(l)]
    exp = [l for l in c["lines"] if not _volatile(l)]
    assert got == exp


def test_cli_fails_loudly_without_gpu_or_db(tmp_path):
    """No GPU here / no database: the tool reports 'ABORT' and a non-zero exit code, never an empty result file."""
    build.build_library()
    r = subprocess.run([build.MCQ, "query", str(tmp_path / "nodb"), os.path.join(GOLD, "cli_reads.fa")], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "ABORT" in r.stderr
    r = subprocess.run([build.MCQ, "query", "toy32", "cli_reads.fa", "-bogus"], cwd=GOLD, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "unknown option" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["default", "hitdiff_percent", "pairseq_insert", "mapped_only_vote", "hits_per_ref_lineage", "two_files", "precision"])
def test_cli_resident_parts_single_part_equals_reference(tmp_path, case):
    """-resident-parts / -gpus (mc_partset_*: parts as contexts of their own, candidates gathered over RCCL, merged on the device): on
    the single-part toy database the output must be the reference's, line by line."""
    build.build_library()
    c = CASES[case]
    out = tmp_path / "out.txt"
    cmd = [build.MCQ, "query", "toy32"] + c["files"] + c["args"] + ["-resident-parts", "1", "-gpus", "0", "-out", str(out)]
    r = subprocess.run(cmd, cwd=GOLD, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    unordered = "hits_per_ref" in case
    got = [l for l in out.read_text().split("\n") if not _volatile(l) and "threads" not in l]
    exp = [l for l in c["lines"] if not _volatile(l) and "threads" not in l]
    assert (sorted(got) == sorted(exp)) if unordered else (got == exp)


@pytest.mark.gpu
@pytest.mark.parametrize("case,shards", [("default", 3), ("pairseq_insert", 2), ("mapped_only_vote", 4), ("two_files", 3), ("precision", 1)])
def test_cli_key_shards_equal_reference(tmp_path, case, shards):
    """-shard keys (mc_keyset_*: ONE database key-sharded, every shard looks up the features it owns, partial location lists as 4-byte
    numbers to the read's owner, candidates there): the output must be the reference's, line by line.  One GPU: the shards share it;
    shards = 1 runs the exchange through RCCL (a rank that sends to itself)."""
    build.build_library()
    c = CASES[case]
    out = tmp_path / "out.txt"
    cmd = [build.MCQ, "query", "toy32"] + c["files"] + c["args"] + ["-shard", "keys", "-key-shards", str(shards), "-gpus", "0", "-out", str(out)]
    env = dict(os.environ, MC_KEYSET_RCCL="1") if shards == 1 else None
    r = subprocess.run(cmd, cwd=GOLD, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr
    got = [l for l in out.read_text().split("\n") if not _volatile(l) and "threads" not in l]
    exp = [l for l in c["lines"] if not _volatile(l) and "threads" not in l]
    assert got == exp


@pytest.mark.gpu
@pytest.mark.parametrize("case,ranges,resident", [("default", 3, 0), ("pairseq_insert", 2, 1), ("mapped_only_vote", 5, 2), ("two_files", 4, 0), ("hitdiff_percent", 40, 7)])
def test_cli_target_ranges_equal_reference(tmp_path, case, ranges, resident):
    """-shard targets (mc_config.target_shard_*: ONE database file cut into contiguous target ranges at load, every range a context of
    its own, the ranges' candidates merged in range order): the output must be the reference's, line by line.  One GPU: the ranges share
    it; -resident-parts: that many ranges in HBM at a time."""
    build.build_library()
    c = CASES[case]
    out = tmp_path / "out.txt"
    cmd = [build.MCQ, "query", "toy32"] + c["files"] + c["args"] + ["-shard", "targets", "-target-shards", str(ranges), "-gpus", "0", "-out", str(out)]
    if resident:
        cmd += ["-resident-parts", str(resident)]
    r = subprocess.run(cmd, cwd=GOLD, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    got = [l for l in out.read_text().split("\n") if not _volatile(l) and "threads" not in l]
    exp = [l for l in c["lines"] if not _volatile(l) and "threads" not in l]
    assert got == exp


@pytest.mark.gpu
@pytest.mark.parametrize("resident", [1, 2, 3])
def test_cli_resident_parts_equal_all_parts_resident(tmp_path, resident):
    """the four-part fixture, `resident` parts in HBM at a time: the same output as with every part in one table"""
    build.build_library()
    args = ["cli_reads.fa", "cli_pairs.fq", "-tophits", "-queryids", "-lowest", "species", "-maxcand", "3"]
    outs = []
    env = dict(os.environ, MC_PARTSET_RCCL="1") if resident == 2 else None     # once through RCCL from the stand-alone program (25 s of RCCL start-up)
    for extra in ([], ["-resident-parts", str(resident)]):
        out = tmp_path / f"out{len(outs)}.txt"
        r = subprocess.run([build.MCQ, "query", "toy32p4"] + args + extra + ["-out", str(out)], cwd=GOLD, capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr
        outs.append([l for l in out.read_text().split("\n") if not _volatile(l) and "threads" not in l])
    assert outs[0] == outs[1] and len(outs[0]) > 500
