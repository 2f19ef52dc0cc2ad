"""`build` wall time from FASTA files on disk: mcq (this repo, MI355X) beside the reference's own `metacache build` (CPU, all host
threads) on the same files.  N unrelated genomes of L bases, one 80-column FASTA file each, taxon id in the header.

    python tools/build_bench.py [--genomes 200] [--length 5000000] [--out gpurun_out/build.json]
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from metacache_amd import build  # noqa: E402


def run(cmd):
    t0 = time.perf_counter()
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(" ".join(cmd[:4]) + "\n" + r.stderr[-2000:])
    return time.perf_counter() - t0, r.stdout


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genomes", type=int, default=200)
    ap.add_argument("--length", type=int, default=5_000_000)
    ap.add_argument("--out", default="")
    ap.add_argument("--no-ref", action="store_true")
    args = ap.parse_args()
    build.build_library()
    tmp = tempfile.mkdtemp(prefix="mcbuild", dir="/tmp")
    G, GL = args.genomes, args.length
    rng = np.random.default_rng(16)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    files = []
    rows = GL // 80 + 1
    for i in range(G):
        g = acgt[rng.integers(0, 4, size=rows * 80, dtype=np.uint8)]
        body = np.full((rows, 81), ord("\n"), dtype=np.uint8)
        body[:, :80] = g.reshape(rows, 80)
        fn = os.path.join(tmp, f"syn{i:04d}.fa")
        with open(fn, "wb") as f:
            f.write(f">SYN_{i:06d}.1 synthetic genome taxid|{1000 + i}|\n".encode())
            f.write(body.tobytes())
        files.append(fn)
    bases = G * rows * 80
    res = {"genomes": G, "bases": bases, "fasta_bytes": sum(os.path.getsize(f) for f in files), "host_threads": os.cpu_count()}
    run([build.MCQ, "build", os.path.join(tmp, "warm")] + files[:1] + ["-silent"])
    wall, out = run([build.MCQ, "build", os.path.join(tmp, "mcqdb")] + files)
    times = {l.split(":")[0].strip(): float(l.split(":")[1].split()[0]) for l in out.splitlines() if l.startswith(("Construction time", "Writing time"))}
    res["mcq_build"] = {"wall_s": round(wall, 3), "Mbp_per_s": round(bases / 1e6 / wall, 1), "construction_s": times.get("Construction time"),
                        "writing_s": times.get("Writing time"), "db_bytes": os.path.getsize(os.path.join(tmp, "mcqdb.cache0"))}
    ref = os.path.join(ROOT, "oracle", "_ref", "metacache_u32")
    if os.path.exists(ref) and not args.no_ref:
        wall, out = run([ref, "build", os.path.join(tmp, "refdb")] + files)
        times = {l.split(":")[0].strip(): float(l.split(":")[1].split()[0]) for l in out.splitlines() if l.startswith(("Construction time", "Writing time"))}
        res["reference_cpu_build"] = {"wall_s": round(wall, 3), "Mbp_per_s": round(bases / 1e6 / wall, 1), "construction_s": times.get("Construction time"),
                                      "writing_s": times.get("Writing time"), "db_bytes": os.path.getsize(os.path.join(tmp, "refdb.cache0"))}
    print(json.dumps(res, indent=1))
    if args.out:
        os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(res, f, indent=1)
    for f in os.listdir(tmp):
        os.remove(os.path.join(tmp, f))


if __name__ == "__main__":
    main()
