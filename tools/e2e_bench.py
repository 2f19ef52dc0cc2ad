"""End-to-end `query` wall time ("kernel-only AND end-to-end"): FASTA file on disk -> per-read classification
output, `mcq` (this repo, MI355X) beside the reference's own CLI (oracle/_ref/metacache_u16, CPU, all host threads) on the
SAME database files and the SAME read file; their `-tophits -queryids` mapping lines are diffed (sorted: the reference's
multi-threaded output order is not deterministic).  Never bench.py's `value` -- this includes parsing, PCIe and printing.

    python tools/e2e_bench.py [--reads 4000000] [--out gpurun_out/e2e.json]
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
import torch  # noqa: E402
import bench  # noqa: E402
from metacache_amd import api, build  # noqa: E402


def write_fasta(path: str, reads: np.ndarray):
    """reads [n, 150] uint8 -> '>r00000000 synthetic\\n' + bases + '\\n' per record, vectorised"""
    n, L = reads.shape
    hdr = np.frombuffer(b">r00000000 synthetic\n", dtype=np.uint8)
    rec = np.empty((n, hdr.size + L + 1), dtype=np.uint8)
    rec[:, : hdr.size] = hdr
    idx = np.arange(n, dtype=np.int64)
    for d in range(8):
        rec[:, 2 + 7 - d] = ord("0") + (idx // 10 ** d) % 10
    rec[:, hdr.size: hdr.size + L] = reads
    rec[:, -1] = ord("\n")
    rec.tofile(path)


def speed_of(path: str):
    q = t = None
    with open(path) as f:
        for line in f:
            if line.startswith("# queries:"):
                q = int(line.split()[2])
            elif line.startswith("# time:"):
                t = float(line.split()[2])
    return q, t


def run(cmd, **kw):
    t0 = time.perf_counter()
    r = subprocess.run(cmd, capture_output=True, text=True, **kw)
    for line in r.stderr.splitlines():
        if line.startswith("mcq profile"):
            print(" ".join(cmd[4:8]), "|", line, flush=True)
    if r.returncode != 0:
        raise RuntimeError(" ".join(cmd) + "\n" + r.stderr[-2000:])
    return time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=4_000_000)
    ap.add_argument("--cpu-reads", type=int, default=400_000, help="reads whose mapping lines are diffed against the reference CLI")
    ap.add_argument("--out", default="")
    ap.add_argument("--no-ref", action="store_true")
    ap.add_argument("--build", action="store_true", help="also time `build` from FASTA files (mcq and the reference)")
    args = ap.parse_args()
    build.build_library()
    G, GL = 16, 5_000_000
    tmp = tempfile.mkdtemp(prefix="mce2e", dir="/tmp")
    genomes = bench.make_genomes(G, GL, 16)
    bld = api.Builder(target_id_bytes=2, max_candidates=2)
    for i, g in enumerate(genomes):
        bld.add_target(g, f"SYN_{i:06d}.1", parent_taxid=1000 + i, filename=f"syn{i}.fa")
    bld.finish(load=False)
    taxa = [(1, 1, 20, "root")] + [(1000 + i, 1, 4, f"synthetic species {i}") for i in range(G)]
    db = os.path.join(tmp, "syn16")
    bld.write(db, taxa)
    bld.free()
    dev = torch.device("cuda", 0)
    gcat = torch.from_numpy(np.concatenate(genomes)).to(dev)
    goff = torch.arange(G, device=dev, dtype=torch.int64) * GL
    parts = [bench.synth_reads_gpu(gcat, goff, GL, 1_000_000, 1016 + i)[:, :150].cpu().numpy() for i in range((args.reads + 999_999) // 1_000_000)]
    reads = np.concatenate(parts)[: args.reads]
    fa, fa_small = os.path.join(tmp, "reads.fa"), os.path.join(tmp, "reads_small.fa")
    write_fasta(fa, reads)
    write_fasta(fa_small, reads[: args.cpu_reads])
    del gcat, parts, reads
    torch.cuda.empty_cache()

    res = {"reads": args.reads, "cpu_reads": args.cpu_reads, "host_threads": os.cpu_count(), "fasta_bytes": os.path.getsize(fa)}
    mcq = build.MCQ
    ref = os.path.join(ROOT, "oracle", "_ref", "metacache_u16")
    o = os.path.join(tmp, "o.txt")
    run([mcq, "query", db, fa_small, "-no-map", "-out", o])                               # page cache + first-touch warm-up
    for name, extra in (("mcq_nomap", ["-no-map"]), ("mcq_map", []), ("mcq_tophits_ids", ["-tophits", "-queryids"])):
        wall = run([mcq, "query", db, fa] + extra + ["-out", o])
        q, ms = speed_of(o)
        res[name] = {"wall_s_incl_db_load": round(wall, 3), "query_ms": ms, "Mreads_per_min": round(q / (ms / 1e3) * 60 / 1e6, 1)}
    for t in (1, 8, 32, 64, 128):
        wall = run([mcq, "query", db, fa, "-no-map", "-threads", str(t), "-out", o])
        q, ms = speed_of(o)
        res[f"mcq_nomap_threads{t}"] = {"query_ms": ms, "Mreads_per_min": round(q / (ms / 1e3) * 60 / 1e6, 1)}
    for t, bs in ((16, 262144), (16, 1048576)):
        wall = run([mcq, "query", db, fa, "-no-map", "-threads", str(t), "-batch-size", str(bs), "-out", o])
        q, ms = speed_of(o)
        res[f"mcq_nomap_threads{t}_batch{bs}"] = {"query_ms": ms, "Mreads_per_min": round(q / (ms / 1e3) * 60 / 1e6, 1)}
    if os.path.exists(ref) and not args.no_ref:
        oref = os.path.join(tmp, "oref.txt")
        wall = run([ref, "query", db, fa, "-no-map", "-out", oref])
        q, ms = speed_of(oref)
        res["reference_cpu_nomap"] = {"wall_s_incl_db_load": round(wall, 3), "query_ms": ms, "threads": os.cpu_count(),
                                      "Mreads_per_min": round(q / (ms / 1e3) * 60 / 1e6, 2)}
        run([ref, "query", db, fa_small, "-tophits", "-queryids", "-out", oref])
        run([mcq, "query", db, fa_small, "-tophits", "-queryids", "-out", o])
        a = sorted(l for l in open(oref) if not l.startswith("#"))
        b = sorted(l for l in open(o) if not l.startswith("#"))
        res["identical_mapping_lines"] = {"reference": len(a), "mcq": len(b), "differing": sum(x != y for x, y in zip(a, b)) + abs(len(a) - len(b))}
    if args.build:
        # `build` from FASTA files on disk (80-column lines, one file per genome, taxon id in the header): mcq beside the reference
        gdir = os.path.join(tmp, "genomes")
        os.makedirs(gdir)
        files = []
        for i, g in enumerate(genomes):
            fn = os.path.join(gdir, f"syn{i:03d}.fa")
            body = np.full((GL // 80 + 1, 81), ord("\n"), dtype=np.uint8)
            pad = np.full(body.shape[0] * 80, ord("A"), dtype=np.uint8)
            pad[:GL] = g
            body[:, :80] = pad.reshape(-1, 80)
            with open(fn, "wb") as f:
                f.write(f">SYN_{i:06d}.1 synthetic genome taxid|{1000 + i}|\n".encode())
                f.write(body.tobytes()[: GL + GL // 80 + 1])
                f.write(b"\n")
            files.append(fn)
        bdb = os.path.join(tmp, "built_mcq")
        run([mcq, "build", bdb] + files[:1] + ["-silent"])                      # warm-up (runtime start, page cache)
        wall = run([mcq, "build", bdb] + files + ["-silent"])
        res["mcq_build"] = {"wall_s": round(wall, 3), "Mbp": G * GL / 1e6, "Mbp_per_s": round(G * GL / 1e6 / wall, 1)}
        ref32 = os.path.join(ROOT, "oracle", "_ref", "metacache_u32")
        if os.path.exists(ref32) and not args.no_ref:
            rdb = os.path.join(tmp, "built_ref")
            wall = run([ref32, "build", rdb] + files + ["-silent"])
            res["reference_cpu_build"] = {"wall_s": round(wall, 3), "threads": os.cpu_count(), "Mbp_per_s": round(G * GL / 1e6 / wall, 1)}
    print(json.dumps(res, indent=1))
    if args.out:
        os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
