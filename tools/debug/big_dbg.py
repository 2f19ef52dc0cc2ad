import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import cpuref, scale_util
from metacache_amd import api, synth, synthdb

lowest, K = int(sys.argv[1]), int(sys.argv[2])
spec = synthdb.phylogeny(120, 2, 3, 40_000, 60_000, seed=100 + lowest + K)
sk = dict(kmerlen=10, sketchlen=16, winlen=121, winstride=112)
cs = synthdb.CpuSynth()
P = synthdb.read_params(spec, 77, sub_rate=0.02)
reads = [bytes(r[:150]) for r in cs.reads(spec, P, 0, 2500)]
rng = np.random.default_rng(5)
reads += [bytes(synth.random_genome(rng, 150)) for _ in range(700)]
reads += [r[:70] for r in reads[:300]]
odb = scale_util.oracle_database(spec, None, threads=64, with_lineages=True, k=10, s=16, w=121, stride=112)
res = {}
for big in ("0", "100000000"):
    os.environ["MC_BIG_MIN"] = big
    db, info = synthdb.build_database(spec, shards=2, max_candidates=K, **sk)
    db.set_lineages(spec.lineages())
    cands, counts, _ = db.query(reads, lowest=lowest)
    res[big] = (cands.copy(), counts.copy())
    db.close()
a, b = res["0"][0], res["100000000"][0]
diff = [i for i in range(len(reads)) if not np.array_equal(a[i], b[i])]
print("queries that differ between big and no-big:", len(diff), diff[:20])
nbad = {"0": 0, "100000000": 0}
for i, r in enumerate(reads):
    h, e = odb.query(r, b"", K, lowest, 0)
    e = e[:K]
    for big in nbad:
        g = res[big][0][i]
        ok = all((g[j]["tgt"], g[j]["hits"], g[j]["beg"], g[j]["end"]) == (e[j]["tgt"], e[j]["hits"], e[j]["beg"], e[j]["end"]) if j < len(e) else g[j]["hits"] == 0 for j in range(K))
        if not ok:
            nbad[big] += 1
            if nbad[big] <= 4:
                tg = int(g[0]["tgt"])
                sel = h[h["tgt"] == tg]
                print("BAD big=", big, "read", i, "H", len(h), "gpu", g, "oracle", e, "oracle hits on gpu's top target:", sel[:20], "species of gpu tgt", spec.species[tg] if tg < len(spec.species) else None,
                      "species of oracle tgts", [int(spec.species[int(x["tgt"])]) for x in e])
print("mismatches vs oracle:", nbad)
