"""Time of mc_build_remove_ambiguous (-remove-ambig-features) on a synthetic build: --genomes related genomes of --length bp in
--taxa taxa; prints one JSON line.  GPU box only."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metacache_amd import api


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genomes", type=int, default=200)
    ap.add_argument("--length", type=int, default=5_000_000)
    ap.add_argument("--taxa", type=int, default=50)
    a = ap.parse_args()
    rng = np.random.default_rng(1)
    base = rng.integers(0, 4, a.length, dtype=np.uint8)
    bld = api.Builder(target_id_bytes=4)
    t0 = time.perf_counter()
    for i in range(a.genomes):
        g = base.copy()
        if i % 4:                                              # three of four genomes: 2 % substitutions of a shared ancestor
            pos = rng.integers(0, a.length, a.length // 50)
            g[pos] = rng.integers(0, 4, pos.size, dtype=np.uint8)
        else:
            g = rng.integers(0, 4, a.length, dtype=np.uint8)
        bld.add_target(np.frombuffer(b"ACGT", dtype=np.uint8)[g], f"G{i}", 0, "f.fa")
    bld.finish(load=False)
    t1 = time.perf_counter()
    keys, vals = bld.counts()
    anc = (np.arange(a.genomes, dtype=np.uint32) % a.taxa) + 1
    rem = bld.remove_ambiguous(anc, 1)
    t2 = time.perf_counter()
    k2, v2 = bld.counts()
    print(json.dumps({"genomes": a.genomes, "length": a.length, "taxa": a.taxa, "build_s": round(t1 - t0, 3), "features": keys,
                      "locations": vals, "removed": rem, "features_after": k2, "locations_after": v2,
                      "remove_ambiguous_ms": round((t2 - t1) * 1e3, 2)}))
    bld.free()


if __name__ == "__main__":
    main()
