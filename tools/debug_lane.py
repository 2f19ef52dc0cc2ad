"""debug: lane path vs wave path on bench-like data"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from metacache_amd import api, synth
import bench

G, GL, B = 16, 5_000_000, 200_000
genomes = bench.make_genomes(G, GL, 16)
def build():
    bld = api.Builder(target_id_bytes=2, max_candidates=2)
    for i, g in enumerate(genomes):
        bld.add_target(g, f"S{i}", 1000 + i)
    db = bld.finish(load=True); bld.free(); return db
dev = torch.device("cuda", 0)
gcat = torch.from_numpy(np.concatenate(genomes)).to(dev)
goff = torch.arange(G, device=dev, dtype=torch.int64) * GL
reads = torch.cat([bench.synth_reads_gpu(gcat, goff, GL, B, 1016).reshape(-1), torch.zeros(16, dtype=torch.uint8, device=dev)])
qinfo = torch.zeros((B, 4), dtype=torch.int32, device=dev)
qinfo[:, 0] = torch.arange(B, device=dev, dtype=torch.int32) * 152; qinfo[:, 1] = 150; qinfo[:, 2] = qinfo[:, 0]
outs = {}
for mode in ("1", "0"):
    os.environ["MC_NO_LANE_PATH"] = mode
    db = build()
    out = torch.zeros((B, 2, 4), dtype=torch.int32, device=dev)
    res = db.query_device(reads.data_ptr(), qinfo.data_ptr(), B, B * 152, max_win_uniform=3)
    db.copy_results(out.data_ptr(), res.cands, B * 2 * 16)
    qs = torch.zeros((B, 4), dtype=torch.int32, device=dev)
    db.copy_results(qs.data_ptr(), res.hit_counts, B * 16)
    db.synchronize()
    outs[mode] = (out.cpu().numpy().view(np.uint32), qs.cpu().numpy().view(np.uint32))
    db.close()
a, qa = outs["1"]; b, qb = outs["0"]
bad = np.nonzero((a != b).any(axis=(1, 2)))[0]
print("mismatching queries:", len(bad), "of", B)
print("qstat sums wave", qa.sum(axis=0), "lane", qb.sum(axis=0), "zero rows wave", (qa.sum(axis=1)==0).sum(), "lane", (qb.sum(axis=1)==0).sum())
print("qstat equal hits:", (qa[:, 0] == qb[:, 0]).mean(), "nfeat:", (qa[:, 1] == qb[:, 1]).mean(), "nfound:", (qa[:, 2] == qb[:, 2]).mean())
for i in bad[:12]:
    print(i, "wave:", a[i].tolist(), "lane:", b[i].tolist(), "qs wave", qa[i].tolist(), "lane", qb[i].tolist())
z = np.nonzero(qa.sum(axis=1) == 0)[0]
print("zero rows (wave) first 40:", z[:40].tolist())
print("zero rows mod 64 histogram:", np.bincount(z % 64, minlength=64).tolist())
