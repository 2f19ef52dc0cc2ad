"""GPU: bounded randomised soak (tools/soak.py): 20 random databases (unrelated genomes + strain groups of 1..40 members at 0..5 %
divergence, repeats, both target widths, load factors 0.3..0.8) x 3 000 random reads each (16..600 bp, mates, N runs, lower case,
K = 1..4, sequence / species / genus level, insert sizes, all list-length classes incl. the filtered big-list kernel) against the C
oracle, candidate for candidate.  The log goes to gpurun_out/ (copied to profiles/ per round)."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bounded_soak_against_oracle():
    spec = importlib.util.spec_from_file_location("soak", os.path.join(ROOT, "tools", "soak.py"))
    soak = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(soak)
    lines = []
    total, bad = soak.run(iters=20, seed=2, nreads=3000, log=lines.append)
    lines.append(f"SOAK {'OK' if bad == 0 else 'FAILED'} {total} queries, {bad} mismatches")
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "soak.log"), "w") as f:
        f.write("\n".join(lines) + "\n")
    assert total == 60000 and bad == 0, lines[-5:]
