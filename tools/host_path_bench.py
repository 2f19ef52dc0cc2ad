"""PCIe-inclusive rate of the host slot path (never bench.py's `value`): reads start in ordinary host memory,
mc_batch_add_bulk copies them into the slot's pinned buffer, submit = H2D + kernels + D2H of the candidates."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from metacache_amd import api

G, GL, N = 16, 5_000_000, 4_000_000
genomes = bench.make_genomes(G, GL, 16)
bld = api.Builder(target_id_bytes=2, max_candidates=2, slot_max_queries=1 << 20, slot_max_chars=(1 << 20) * 152 + 64)
for i, g in enumerate(genomes):
    bld.add_target(g, f"S{i}", 1000 + i)
db = bld.finish(load=True); bld.free()
dev = torch.device("cuda", 0)
gcat = torch.from_numpy(np.concatenate(genomes)).to(dev)
goff = torch.arange(G, device=dev, dtype=torch.int64) * GL
reads = torch.cat([bench.synth_reads_gpu(gcat, goff, GL, 1_000_000, 1016 + i)[:, :150] for i in range(N // 1_000_000)]).cpu().numpy()
seqs = np.ascontiguousarray(reads).reshape(-1)
offs = np.arange(N + 1, dtype=np.uint64) * np.uint64(150)
db.query_bulk(seqs[: 150 * 100000], offs[:100001])          # warm up
t0 = time.perf_counter()
c = db.query_bulk(seqs, offs)
t = time.perf_counter() - t0
print(f"host slot path, 1 thread, 1 slot: {N} reads in {t:.3f} s = {N / t * 60 / 1e6:.0f} Mreads/min (PCIe + host copy inclusive); "
      f"classified {(c['hits'][:, 0] > 0).mean():.3f}")
db.close()
