"""The C++ mirror of the reference's query interface (include/metacache_amd.hpp): compiles and links with plain g++ on
CPU; on the GPU the example program (the reference's query_gpu loop) must print what the Python binding returns."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "query_example")


def build_example():
    from metacache_amd import build
    build.build_library()
    cmd = ["g++", "-std=c++14", "-Wall", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "query_example.cpp"),
           "-L" + os.path.join(ROOT, "metacache_amd", "lib"), "-lmetacache_amd", "-Wl,-rpath," + os.path.join(ROOT, "metacache_amd", "lib"),
           "-L/opt/rocm/lib", "-Wl,-rpath-link,/opt/rocm/lib", "-o", EXE]
    subprocess.check_call(cmd)
    return EXE


def test_cpp_mirror_compiles_and_links():
    assert os.path.exists(build_example())


@pytest.mark.gpu
def test_cpp_example_matches_python_binding(golden, tmp_path):
    from metacache_amd import api
    exe = build_example()
    single, _, _ = golden.reads()
    reads = [r for r in single[:400] if b"\n" not in r and len(r) > 0]
    f = tmp_path / "seqs.txt"
    f.write_bytes(b"\n".join(reads) + b"\n")
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    out = subprocess.check_output([exe, golden.db_path("toy32"), str(f), "4"], env=env).decode().splitlines()
    db = api.Database.open(golden.db_path("toy32"), max_candidates=2)
    cands, _, _ = db.query(reads, lowest=4)
    db.close()
    assert len(out) == len(reads)
    for i, line in enumerate(out):
        idx, rest = line.split("\t")
        exp = "".join(f"{c['tgt']}:{c['hits']}:{c['beg']}-{c['end']}," for c in cands[i] if c["hits"] > 0)
        assert int(idx) == i and rest == exp, (i, rest, exp)
