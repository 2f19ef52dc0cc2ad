"""Modes R / P / K of metacache_amd/distributed.py over a real process group (RCCL), every rank on its own GPU, checked against the
oracle on rank 0.  Run directly (world size 1) or under torch.distributed.run --nproc-per-node N.  TEST TOOL (loads the oracle)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cpuref  # noqa: E402
from metacache_amd import api  # noqa: E402
from metacache_amd.distributed import (classify_key_sharded_device, classify_partitioned, classify_sharded, gather_candidates,  # noqa: E402
                                       shard_bounds)


def device_batch(reads, dev):
    pad = [len(r) + (-len(r)) % 4 for r in reads]
    offs = np.concatenate([[0], np.cumsum(pad)]).astype(np.int64)
    buf = np.zeros(int(offs[-1]) + 16, dtype=np.uint8)
    for r, o in zip(reads, offs[:-1]):
        buf[o:o + len(r)] = np.frombuffer(r, dtype=np.uint8)
    qinfo = np.zeros((len(reads), 4), dtype=np.uint32)
    qinfo[:, 0] = offs[:-1]; qinfo[:, 1] = [len(r) for r in reads]; qinfo[:, 2] = offs[:-1]
    return torch.from_numpy(buf).to(dev), torch.from_numpy(qinfo.view(np.int32)).to(dev), int(offs[-1])


def main():
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    gold = os.path.join(ROOT, "tests", "golden")
    z = np.load(os.path.join(gold, "toy_reads.npz"))
    off = z["single_off"]
    reads = [z["single"][int(off[i]):int(off[i + 1])].tobytes() for i in range(1500)]
    reads = [r for r in reads if len(r) == 150]                       # uniform maxWindowsInRange
    n, K = len(reads), 2
    seq, qinfo, nchars = device_batch(reads, dev)
    orc = cpuref.oracle()
    bad = 0

    def compare(got, odb, lowest, mode=0):
        nonlocal bad
        g = got.cpu().numpy().view(np.uint32)
        for i in range(n):
            _, e = odb.query(reads[i], b"", K, lowest, 0, mode=mode)
            for j in range(K):
                exp = (int(e[j]["tgt"]), int(e[j]["hits"]), int(e[j]["beg"]), int(e[j]["end"])) if j < len(e) else None
                if exp is None:
                    bad += int(g[i, j, 1] != 0)
                else:
                    bad += int(tuple(int(x) for x in g[i, j]) != exp)

    # ---- Mode R: replicated table, reads sharded, gather to rank 0
    db = api.Database.open(os.path.join(gold, "toy32"), device=local, max_candidates=K)
    mw = db.max_windows_in_range(150)

    def classify(lo, hi):
        s, q, nc = device_batch(reads[lo:hi], dev)
        res = db.query_device(s.data_ptr(), q.data_ptr(), hi - lo, nc, max_win_uniform=mw)
        out = torch.empty((hi - lo, K, 4), dtype=torch.int32, device=dev)
        db.copy_results(out.data_ptr(), res.cands, (hi - lo) * K * 16); db.synchronize()
        return out
    got = classify_sharded(n, classify)
    odb = orc.open(os.path.join(gold, "toy32"))
    if rank == 0:
        compare(got, odb, 0)
    db.close()
    # ---- Mode K: this rank holds the features it owns; partial lists -> all-to-all -> union kernel -> candidates
    for lowest in (0, 4):
        dbk = api.Database.open(os.path.join(gold, "toy32"), device=local, max_candidates=K, key_shard_index=rank, key_shard_count=world)
        for wire in (8, 4):                                        # (target, window) pairs, then 4-byte global window numbers
            res = dbk.query_device(seq.data_ptr(), qinfo.data_ptr(), n, nchars, max_win_uniform=mw, want_partial_hits=(lowest == 0 and wire == 8),
                                   want_partial_numbers=(lowest == 0 and wire == 4), want_allhits=(lowest != 0))
            local_c = classify_key_sharded_device(dbk, res, n, K, mw, lowest=lowest, wire=wire)
            parts = gather_candidates(local_c)
            if rank == 0:
                compare(torch.cat(parts, dim=0), odb, lowest)
        dbk.close()
    odb.close()
    # ---- Mode P: rank r holds part r % 2 of the 2-part database (world 1: both parts one after the other), merge per read
    owhole = orc.open(os.path.join(gold, "toy32p2"))
    for lowest in (0, 4):
        per_part, per_tax = [], []
        my_parts = [rank % 2] if world > 1 else [0, 1]
        for p in my_parts:
            dbp = api.Database.open(os.path.join(gold, "toy32p2"), device=local, max_candidates=K, single_part=p)
            res = dbp.query_device(seq.data_ptr(), qinfo.data_ptr(), n, nchars, max_win_uniform=mw, lowest=lowest)
            c = torch.empty((n, K, 4), dtype=torch.int32, device=dev)
            dbp.copy_results(c.data_ptr(), res.cands, n * K * 16); dbp.synchronize()
            lin = torch.from_numpy(dbp.lineages().astype(np.int64)).to(dev)
            tg = c[:, :, 0].to(torch.int64).clamp(min=0, max=lin.shape[0] - 1)
            if lowest:
                col = lin[:, lowest:]
                first = (col != 0).to(torch.int64).argmax(dim=1)
                taxkey = col.gather(1, first[:, None])[:, 0]
            else:
                taxkey = torch.arange(lin.shape[0], device=dev) + 1
            tx = torch.where(c[:, :, 1] > 0, taxkey[tg], torch.zeros_like(tg))
            per_part.append(c); per_tax.append(tx)
            dbp.close()
        from metacache_amd.distributed import merge_part_candidates
        if world == 1:
            merged = merge_part_candidates(per_part, per_tax if lowest else None)
        else:
            merged = classify_partitioned(per_part[0], taxa=per_tax[0] if lowest else None)
        if rank == 0 and world <= 2:
            compare(merged, owhole, lowest, mode=1)
    owhole.close()
    t = torch.tensor([bad], device=dev)
    dist.all_reduce(t)
    dist.barrier()
    if rank == 0:
        print("DIST MODES OK" if int(t.item()) == 0 else f"DIST MODES FAILED: {int(t.item())} mismatches", flush=True)
    dist.destroy_process_group()
    sys.exit(0 if int(t.item()) == 0 else 1)


if __name__ == "__main__":
    main()
