#!/usr/bin/env python3
"""Golden vectors for `merge`: result files of the reference queried against ONE database part at a time
(`metacache query toy32p2.cache<i> ... -tophits -queryids -lowest species`, docs/partitioning.md:116-153) and what the reference's
`metacache merge` makes of them.  Runs only in the build container (needs oracle/_ref/metacache_u32).  Writes data only:

  merge_in/part0.txt, part1.txt, part0_genus.txt, part1_genus.txt     per-part result files
  merge_expected.json.gz                                              {case: {"args": [...], "lines": [...]}}

Usage:  python tests/golden/make_golden_merge.py
"""
import gzip
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref", "metacache_u32")
TAX = "build_in/taxonomy"

CASES = {
    "default": (["merge_in/part0.txt", "merge_in/part1.txt"], []),
    "species_lineage": (["merge_in/part1.txt", "merge_in/part0.txt"], ["-lowest", "species", "-lineage", "-taxids", "-maxcand", "3", "-tophits"]),
    "genus": (["merge_in/part0_genus.txt", "merge_in/part1_genus.txt"], ["-lowest", "genus", "-highest", "family", "-tophits", "-queryids"]),
    "vote": (["merge_in/part0.txt", "merge_in/part1.txt"], ["-hitmin", "3", "-hitdiff", "0.5", "-maxcand", "4", "-tophits", "-mapped-only"]),
    "abundances": (["merge_in/part0.txt", "merge_in/part1.txt"], ["-abundances", "-abundance-per", "genus", "-no-map"]),
    "directory": (["merge_in"], ["-tophits", "-maxcand", "4", "-no-summary"]),
    "separate_cols": (["merge_in/part0.txt", "merge_in/part1.txt"], ["-separate-cols", "-lineage", "-taxids", "-no-query-params"]),
}


def run(cmd):
    r = subprocess.run(cmd, cwd=HERE, capture_output=True, text=True, timeout=600)
    if r.returncode != 0:
        sys.exit(f"FAILED: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
    return r


def main():
    if not os.path.exists(REF):
        sys.exit("oracle/_ref is missing: run `make -C oracle ref` first")
    os.makedirs(os.path.join(HERE, "merge_in"), exist_ok=True)
    for part in (0, 1):
        run([REF, "query", f"toy32p2.cache{part}", "cli_reads.fa", "cli_pairs.fq", "-tophits", "-queryids", "-lowest", "species", "-threads", "1",
             "-out", f"merge_in/part{part}.txt"])
        run([REF, "query", f"toy32p2.cache{part}", "cli_reads.fa", "-tophits", "-queryids", "-lowest", "genus", "-maxcand", "3", "-threads", "1",
             "-out", f"merge_in/part{part}_genus.txt"])
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for name, (files, args) in CASES.items():
            res = os.path.join(tmp, name + ".txt")
            if name == "directory":        # a directory of result files: only the two species-level files may be in it
                d = os.path.join(tmp, "merge_dir")
                os.makedirs(d)
                for part in (0, 1):
                    os.symlink(os.path.join(HERE, f"merge_in/part{part}.txt"), os.path.join(d, f"part{part}.txt"))
                files = [d]
            r = run([REF, "merge"] + files + ["-taxonomy", TAX] + args + ["-out", res])
            out[name] = {"files": CASES[name][0], "args": args, "lines": open(res).read().split("\n")}
    with gzip.open(os.path.join(HERE, "merge_expected.json.gz"), "wt") as f:
        json.dump(out, f)
    print({k: len(v["lines"]) for k, v in out.items()})


if __name__ == "__main__":
    main()
