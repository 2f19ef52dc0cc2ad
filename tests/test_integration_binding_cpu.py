"""INTEGRATION.md sections 1-2 are the binding a maintainer of the reference would add (amd_feature_store, query_amd).  Where the
reference's headers are present (this container; not the GPU box) the C++ blocks of those sections are taken out of the document as
they stand and compiled against database.hpp / candidate_structs.hpp / options.hpp / database_query.hpp and include/metacache_amd.h
(g++ -std=c++14 -fsyntax-only; query_amd is instantiated so that its body is checked too).  Nothing built here travels anywhere."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src"

PRELUDE = """#include <numeric>
#include <iostream>
#include <stdexcept>
#include <unordered_map>
#include "database.hpp"
#include "candidate_structs.hpp"
#include "candidate_generation.hpp"
#include "options.hpp"
#include "database_query.hpp"
#include "io_serialize.hpp"
#include "span.hpp"
using namespace mc;
"""
# the reference's own query_gpu is a template over the buffer and its update functor (database_query.hpp:87-124): instantiate ours alike
INSTANCE = """
struct probe_update { template <class Q, class A, class T> void operator()(int&, const Q&, const A&, const T&) const {} };
template void query_amd<int, probe_update>(mc_ctx*, const database&, const query_options&, const std::vector<sequence_query>&,
                                           unsigned, int&, probe_update&);
void probe_store(amd_feature_store& s, const sketching_opt& sk, const query_options& o, std::istream& is) { s.open(sk, o, 4); s.read_part(is, 0); }
"""


def binding_blocks():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("## 1. Loading"):text.index("## 2b.")]
    return re.findall(r"```cpp\n(.*?)```", sec, re.S)


def test_document_has_the_two_bindings():
    blocks = binding_blocks()
    assert len(blocks) >= 2
    assert "class amd_feature_store" in blocks[0] and "void query_amd(" in blocks[1]


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference's headers are not on this machine")
def test_binding_compiles_against_the_reference_headers(tmp_path):
    blocks = binding_blocks()
    src = tmp_path / "integration_binding.cpp"
    src.write_text(PRELUDE + blocks[0] + blocks[1] + INSTANCE)
    r = subprocess.run(["g++", "-std=c++14", "-fsyntax-only", "-Wall", "-I" + REF, "-I" + os.path.join(ROOT, "include"), str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
