"""GPU: a partitioned database queried part group by part group (mc_partset_*: `resident` parts in HBM at a time, the next group loading
behind the queries, per-part candidates gathered with ncclAllGather and merged on the device in part order) and the merge itself
(mc_merge_part_candidates) -- against the oracle's intended multi-part semantics on the whole file set ("per-part sorted lists
concatenated in part order", host_hashmap.hpp:695-723; the in-process reference is history dependent for more than one part, SURVEY 8a
row 8) and against one context that holds all parts.  Fixture: tests/golden/toy32p4 (make_golden_parts.py)."""
import ctypes as C

import numpy as np
import pytest

import cpuref
from metacache_amd import api

pytestmark = pytest.mark.gpu


def _eq(got, exp, K, tag):
    exp = exp[:K]
    for j in range(K):
        if j < len(exp):
            assert (got[j]["tgt"], got[j]["hits"], got[j]["beg"], got[j]["end"]) == (exp[j]["tgt"], exp[j]["hits"], exp[j]["beg"], exp[j]["end"]), (tag, j, got, exp)
        else:
            assert got[j]["hits"] == 0, (tag, j, got, exp)


@pytest.mark.parametrize("resident,lowest,K", [(2, 0, 2), (2, 4, 3), (1, 0, 2), (3, 4, 2), (4, 0, 4), (0, 6, 2)])
def test_part_groups_equal_intended_multipart(golden, resident, lowest, K, monkeypatch):
    monkeypatch.setenv("MC_PARTSET_RCCL", "1")                   # one GPU: the RCCL calls run with a single rank (several devices: always)
    single, p1, p2 = golden.reads()
    name = golden.db_path("toy32p4")
    odb = cpuref.oracle().open(name)
    ps = api.PartSet(name, resident=resident, max_candidates=K, slot_max_queries=500, slot_max_chars=1 << 17)    # several batches per group
    info = ps.info()
    assert info["parts"] == 4 and info["resident"] == (resident or 4) and info["groups"] == -(-4 // (resident or 4)) and info["devices"] == 1
    got = ps.classify(single, lowest=lowest)
    for i, s in enumerate(single):
        _, c = odb.query(s, b"", K, lowest, 0, mode=1)
        _eq(got[i], c, K, ("single", i))
    gp = ps.classify(p1, p2, lowest=lowest, insert_max=700)
    for i, (a, b) in enumerate(zip(p1, p2)):
        _, c = odb.query(a, b, K, lowest, 700, mode=1)
        _eq(gp[i], c, K, ("pair", i))
    again = ps.classify(single[:300], lowest=lowest)              # a second call starts from the first group again
    assert np.array_equal(again, got[:300])
    ps.close(); odb.close()


def test_part_groups_equal_all_parts_in_one_context(golden):
    single, _, _ = golden.reads()
    name = golden.db_path("toy32p4")
    K = 2
    whole = api.Database.open(name, max_candidates=K)
    cw, _, _ = whole.query(single)
    whole.close()
    ps = api.PartSet(name, resident=2, devices=[0], max_candidates=K)
    got = ps.classify(single)
    ps.close()
    used = cw["hits"] > 0
    for f in ("hits", "beg", "end"):
        assert np.array_equal(got[f], cw[f]), f
    assert np.array_equal(got["tgt"][used], cw["tgt"][used])


@pytest.mark.parametrize("lowest", [0, 4])
def test_merge_kernel_equals_the_python_merge(golden, lowest):
    """mc_merge_part_candidates (one lane per read, the CPU's list insert) against metacache_amd.distributed.merge_part_candidates (with
    and without taxon keys), the torch restatement the gloo tests hold against the oracle"""
    import torch
    from metacache_amd import distributed as D
    single, _, _ = golden.reads()
    reads = single[:900]
    name = golden.db_path("toy32p4")
    K = 3
    parts, keys, dbs = [], [], []
    for p in range(4):
        db = api.Database.open(name, max_candidates=K, single_part=p)
        c, _, _ = db.query(reads, lowest=lowest)
        dbs.append(db)
        t = np.stack([c["tgt"], c["hits"], c["beg"], c["end"]], axis=-1).astype(np.uint32).view(np.int32)
        parts.append(torch.from_numpy(t.copy()).cuda())
    n = len(reads)
    out = torch.zeros((n, K, 4), dtype=torch.int32, device="cuda")
    ptrs = (C.c_void_p * 4)(*[t.data_ptr() for t in parts])
    L = api.lib()
    L.mc_merge_part_candidates.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]
    assert L.mc_merge_part_candidates(dbs[0].h, ptrs, 4, n, lowest, out.data_ptr(), None) == 0
    dbs[0].synchronize()
    if lowest == 0:
        exp = D.merge_part_candidates([t.cpu() for t in parts])
    else:
        lin = dbs[0].lineages()
        taxkey = np.zeros(lin.shape[0], dtype=np.int64)
        for t in range(lin.shape[0]):
            nz = np.flatnonzero(lin[t, lowest:])
            taxkey[t] = lin[t, lowest + nz[0]] if len(nz) else 0
        tk = torch.from_numpy(taxkey)
        pk = [torch.where(t[:, :, 1].cpu() > 0, tk[t[:, :, 0].cpu().long().clamp(min=0, max=len(tk) - 1)], torch.zeros(1, dtype=torch.int64)) for t in parts]
        exp = D.merge_part_candidates([t.cpu() for t in parts], pk)
    g, e = out.cpu().numpy().view(np.uint32), exp.numpy().view(np.uint32)
    live = e[:, :, 1] > 0
    assert np.array_equal(g[:, :, 1:], e[:, :, 1:]) and np.array_equal(g[:, :, 0][live], e[:, :, 0][live])
    for db in dbs:
        db.close()
