"""Mode K on ONE GPU: three key-sharded contexts of the same database stand in for three ranks; the exchange is replaced by
slicing (the collective itself is covered on CPU: tests/test_distributed_cpu.py).  Union of the partial location lists +
mc_candidates_from_hits must equal the unsharded table's result -- and the oracle's -- bit for bit."""
import numpy as np
import pytest
import torch

import cpuref
from metacache_amd import api
from metacache_amd.distributed import shard_bounds, union_partial_hits


def _device_batch(reads, dev):
    pad = [len(r) + (-len(r)) % 4 for r in reads]
    offs = np.concatenate([[0], np.cumsum(pad)]).astype(np.int64)
    buf = np.zeros(int(offs[-1]) + 16, dtype=np.uint8)
    for r, o in zip(reads, offs[:-1]):
        buf[o:o + len(r)] = np.frombuffer(r, dtype=np.uint8)
    qinfo = np.zeros((len(reads), 4), dtype=np.uint32)
    qinfo[:, 0] = offs[:-1]; qinfo[:, 1] = [len(r) for r in reads]; qinfo[:, 2] = offs[:-1]
    return torch.from_numpy(buf).to(dev), torch.from_numpy(qinfo.view(np.int32)).to(dev), int(offs[-1])


@pytest.mark.gpu
@pytest.mark.parametrize("lowest,K", [(0, 2), (4, 3)])
def test_three_key_shards_union_equals_whole_table(golden, lowest, K):
    single, _, _ = golden.reads()
    reads = [r for r in single[:600] + single[1600:1700] if len(r) > 0]
    n = len(reads)
    dev = torch.device("cuda", 0)
    seq, qinfo, nchars = _device_batch(reads, dev)
    whole = api.Database.open(golden.db_path("toy32"), max_candidates=K)
    max_win = torch.tensor([whole.max_windows_in_range(len(r)) for r in reads], dtype=torch.int32, device=dev)
    world = 3
    counts, hits, locs = [], [], 0
    for r in range(world):
        db = api.Database.open(golden.db_path("toy32"), max_candidates=K, key_shard_index=r, key_shard_count=world)
        locs += db.n_locations
        # shard 0 and 2: the sorted lists of the wave path; shard 1: the lane path's lists as they are (MC_WANT_PARTIAL_HITS)
        res = db.query_device(seq.data_ptr(), qinfo.data_ptr(), n, nchars, max_win_ptr=max_win.data_ptr(), want_allhits=(r != 1),
                              want_partial_hits=(r == 1))
        off = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        db.copy_results(off.data_ptr(), res.hit_offsets, (n + 1) * 8)
        db.synchronize()
        h = torch.zeros(int(off[-1]), dtype=torch.int64, device=dev)
        if h.numel():
            db.copy_results(h.data_ptr(), res.hits, h.numel() * 8)
        db.synchronize()
        counts.append(off[1:] - off[:-1]); hits.append((off, h))
        db.close()
    assert locs == whole.n_locations                                  # every feature lives on exactly one shard
    assert all(int(c.sum()) > 0 for c in counts)
    got = torch.zeros((n, K, 4), dtype=torch.int32, device=dev)
    for o in range(world):                                            # owner o: reads [lo, hi)
        lo, hi = shard_bounds(n, o, world)
        psc = torch.stack([c[lo:hi] for c in counts])
        psh = [h[int(off[lo]):int(off[hi])] for off, h in hits]
        offsets, union = union_partial_hits(psc, psh)
        res = whole.candidates_from_hits(union.data_ptr(), offsets.data_ptr(), hi - lo, max_win_ptr=max_win[lo:hi].contiguous().data_ptr(),
                                         lowest=lowest)
        whole.copy_results(got[lo:hi].data_ptr(), res.cands, (hi - lo) * K * 16)
        whole.synchronize()
    # the device data path of the owner side: counts source-major + the sources' pieces back to back -> union kernel + rows 8-10
    got2 = torch.zeros((n, K, 4), dtype=torch.int32, device=dev)
    for o in range(world):
        lo, hi = shard_bounds(n, o, world)
        cnt = torch.cat([c[lo:hi].to(torch.int32) for c in counts]).contiguous()
        pieces = torch.cat([h[int(off[lo]):int(off[hi])] for off, h in hits]).contiguous()
        res = whole.candidates_from_partial_hits(cnt.data_ptr(), pieces.data_ptr() if pieces.numel() else 0, pieces.numel(), hi - lo, world,
                                                 max_win_ptr=max_win[lo:hi].contiguous().data_ptr(), lowest=lowest)
        whole.copy_results(got2[lo:hi].data_ptr(), res.cands, (hi - lo) * K * 16)
        whole.synchronize()
    assert torch.equal(got, got2)
    got = got.cpu().numpy().view(np.uint32)
    exp, _, _ = whole.query(reads, lowest=lowest)
    whole.close()
    odb = cpuref.oracle().open(golden.db_path("toy32"))
    for i in range(n):
        for j in range(K):
            e = exp[i, j]
            assert got[i, j, 1] == e["hits"], (i, j)
            if e["hits"]:
                assert (got[i, j, 0], got[i, j, 2], got[i, j, 3]) == (e["tgt"], e["beg"], e["end"]), (i, j)
        if i % 7 == 0:
            _, c = odb.query(reads[i], b"", K, lowest, 0)
            for j in range(min(K, len(c))):
                assert (got[i, j, 0], got[i, j, 1], got[i, j, 2], got[i, j, 3]) == (c[j]["tgt"], c[j]["hits"], c[j]["beg"], c[j]["end"]), (i, j)
    odb.close()


@pytest.mark.gpu
def test_modes_over_rccl_single_rank_process(golden):
    """tools/dist_modes_check.py in its own process: Mode R gather, Mode P merge (sequence level and per taxon) and the Mode K device
    data path (all-to-all of counts and locations, union kernel) over a real RCCL process group of world size 1 on this GPU, each
    against the oracle.  (World sizes > 1 run the same script under torch.distributed.run; the exchange logic itself is covered with
    two gloo ranks in tests/test_distributed_cpu.py.)"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "dist_modes_check.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "DIST MODES OK" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
