"""Randomised parity soak (development aid; uses the oracle, so it is test infrastructure, not product): random databases
(unrelated genomes + strain groups of random size and divergence, repeats inside genomes, uint16 / uint32 targets), random reads
(single / pairs, 16..600 bp, substitutions, N runs, lower case), random K / lowest rank / insert size / load factor -- HIP path
against the C oracle, candidate for candidate.   python tools/soak.py [--iters 20] [--seed 1]"""
import argparse
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cpuref  # noqa: E402
from metacache_amd import api, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--reads", type=int, default=3000)
    args = ap.parse_args()
    total, bad = run(args.iters, args.seed, args.reads, lambda m: print(m, flush=True))
    print("SOAK", "OK" if bad == 0 else "FAILED", total, "queries,", bad, "mismatches")


def run(iters: int, seed: int, nreads: int, log):
    """-> (queries, mismatches); one log line per database"""
    class A:
        pass
    args = A(); args.iters, args.seed, args.reads = iters, seed, nreads
    orc = cpuref.oracle()
    tmp = tempfile.mkdtemp(prefix="mcsoak", dir="/tmp")
    total = bad = 0
    for it in range(args.iters):
        rng = np.random.default_rng(args.seed * 1000 + it)
        tb = int(rng.choice([2, 4]))
        K = int(rng.integers(1, 5))
        lowest = int(rng.choice([0, 0, 4, 6]))
        lf = float(rng.choice([0.3, 0.5, 0.8]))
        big_min = str(rng.choice([0, 300, 1024]))                   # lists beyond this take big_cands_kernel (read at mc_create)
        os.environ["MC_BIG_MIN"] = big_min
        genomes, parents = [], []
        ngroups = int(rng.integers(2, 8))
        for sp in range(ngroups):
            base = synth.random_genome(rng, int(rng.integers(3000, 40000)))
            if rng.random() < 0.5 and base.size > 6000:
                a, b = int(rng.integers(0, base.size // 2)), int(rng.integers(base.size // 2, base.size - 600))
                L = int(rng.integers(200, 600)); base[b:b + L] = base[a:a + L]
            nst = int(rng.choice([1, 2, 4, 9, 20, 40]))
            div = float(rng.choice([0.0, 0.002, 0.01, 0.05]))
            for st in range(nst):
                genomes.append(synth.mutate(rng, base, div) if st else base)
                parents.append(1000 + sp)
        bld = api.Builder(target_id_bytes=tb, max_candidates=K, max_load_factor=lf)
        for i, g in enumerate(genomes):
            bld.add_target(g, f"S{i:04d}.1", parent_taxid=parents[i])
        name = os.path.join(tmp, f"db{it}")
        bld.finish(load=False)
        bld.write(name, [(1, 1, 20, "root"), (500, 1, 6, "genus a"), (501, 1, 6, "genus b")] +
                  [(1000 + i, 500 + i % 2, 4, f"sp{i}") for i in range(ngroups)])
        bld.free()
        big = [g for g in genomes if g.size >= 700]
        reads, mates = [], []
        for _ in range(args.reads):
            g = big[int(rng.integers(len(big)))]
            L = int(rng.choice([16, 17, 31, 100, 127, 128, 143, 150, 151, 239, 250, 300, 512, 513, 600]))
            p = int(rng.integers(0, g.size - L))
            r = synth.mutate(rng, g[p:p + L], float(rng.choice([0, 0.01, 0.05])))
            if rng.random() < 0.1 and L > 40:
                q = int(rng.integers(0, L - 20)); r[q:q + int(rng.integers(1, 20))] = ord("N")
            if rng.random() < 0.1:
                r = np.frombuffer(bytes(r).lower(), dtype=np.uint8).copy()
            if rng.random() < 0.5:
                r = synth.revcomp(r)
            reads.append(bytes(r))
            L2 = int(rng.choice([0, 0, 50, 120, 150, 300]))
            p2 = min(g.size - L2, p + int(rng.integers(0, 300))) if L2 else 0
            mates.append(bytes(synth.revcomp(g[p2:p2 + L2])) if L2 else b"")
        ins = int(rng.choice([0, 0, 400, 900]))
        odb = orc.open(name)
        db = api.Database.open(name, max_candidates=K, max_load_factor=lf, slot_max_queries=1 << 12, slot_max_chars=1 << 22)
        cands, counts, _ = db.query(reads, mates, lowest=lowest, insert_max=ins)
        db.close()
        nbad = 0
        for i in range(len(reads)):
            _, e = odb.query(reads[i], mates[i], K, lowest, ins)
            e = e[:K]
            ok = all((cands[i, j]["tgt"], cands[i, j]["hits"], cands[i, j]["beg"], cands[i, j]["end"]) ==
                     (e[j]["tgt"], e[j]["hits"], e[j]["beg"], e[j]["end"]) for j in range(len(e))) and \
                all(cands[i, j]["hits"] == 0 for j in range(len(e), K))
            if not ok:
                nbad += 1
                if nbad <= 2:
                    log(f"MISMATCH it {it} read {i} len {len(reads[i])} {len(mates[i])} H {counts[i]} {cands[i]} {e}")
        odb.close()
        total += len(reads); bad += nbad
        log(f"iter {it}: targets {len(genomes)} tb {tb} K {K} lowest {lowest} lf {lf} ins {ins} big_min {big_min} max H {int(counts.max())} "
            f"classes <=24 {int((counts <= 24).sum())} <=256 {int(((counts > 24) & (counts <= 256)).sum())} <=1024 {int(((counts > 256) & (counts <= 1024)).sum())} "
            f">1024 {int((counts > 1024).sum())}: mismatches {nbad}")
        for ext in (".meta", ".cache0"):
            os.remove(name + ext)
    os.environ.pop("MC_BIG_MIN", None)
    return total, bad


if __name__ == "__main__":
    main()
