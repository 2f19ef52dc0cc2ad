#!/usr/bin/env python3
"""Golden vectors for `info` (metadata topics): stdout of the reference's `metacache info <db> [topic ...]` on the toy databases.
Runs only in the build container (oracle/_ref).  Writes tests/golden/info_expected.json.gz.   python tests/golden/make_golden_info.py"""
import gzip
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CASES = {
    "config32": ("metacache_u32", ["toy32"]),
    "config16": ("metacache_u16", ["toy16"]),
    "config_2parts": ("metacache_u32", ["toy32p2"]),
    "targets": ("metacache_u32", ["toy32", "targets"]),
    "targets_named": ("metacache_u32", ["toy32", "target", "NC_000016.1", "NOPE", "NC_000010.1"]),
    "lineages": ("metacache_u32", ["toy32p2", "lineages"]),
    "rank_species": ("metacache_u32", ["toy32", "rank", "species"]),
    "rank_family": ("metacache_u16", ["toy16", "rank", "family"]),
    "basic": ("metacache_u32", []),
}


def main():
    out = {}
    for name, (binary, args) in CASES.items():
        ref = os.path.join(ROOT, "oracle", "_ref", binary)
        if not os.path.exists(ref):
            sys.exit("oracle/_ref is missing: run `make -C oracle ref` first")
        r = subprocess.run([ref, "info"] + args, cwd=HERE, capture_output=True, text=True, timeout=120)
        if r.returncode != 0:
            sys.exit(r.stderr)
        out[name] = {"args": args, "stdout": r.stdout.split("\n")}
    with gzip.open(os.path.join(HERE, "info_expected.json.gz"), "wt") as f:
        json.dump(out, f)
    print({k: len(v["stdout"]) for k, v in out.items()})


if __name__ == "__main__":
    main()
