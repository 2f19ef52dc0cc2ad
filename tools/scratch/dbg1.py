import os, sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from metacache_amd import api
gold = "tests/golden"
z = np.load(os.path.join(gold, "toy_reads.npz"))
off = z["single_off"]
reads = [z["single"][int(off[i]):int(off[i + 1])].tobytes() for i in range(len(off) - 1)]
res = {}
for store in ("0", "1"):
    os.environ["MC_COMPACT_LOCATIONS"] = store
    db = api.Database.open(os.path.join(gold, "toy32"), max_candidates=2, copy_allhits=0)
    print(store, db.table_layout())
    c, counts, _ = db.query(reads)
    res[store] = (c.copy(), counts.copy())
    db.close()
a, b = res["0"], res["1"]
bad = 0
for i in range(len(reads)):
    if not np.array_equal(a[0][i], b[0][i]):
        bad += 1
        if bad <= 12: print(i, len(reads[i]), "H", a[1][i], b[1][i], "wide", a[0][i], "compact", b[0][i])
print("bad", bad, "of", len(reads))
