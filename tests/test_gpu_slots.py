"""GPU: the host slot API under several submitting threads (the reference's consumer threads, database_query.hpp:185-252) -- slots that
are submitted side by side go to the device as ONE batch (the slot coalescer of context.cpp; mc_slot_stats), each slot gets exactly
the candidates it gets alone; pairs, reads of every length class of the goldens, empty slots, a lowest rank above sequence; and the same
with MC_SLOT_COALESCE=0 (every slot its own batch)."""
import ctypes as C
import threading

import numpy as np
import pytest

from metacache_amd import api

pytestmark = pytest.mark.gpu


def _alone(db, reads, mates, lowest):
    c, _, _ = db.query(reads, mates, lowest=lowest)
    return c


def _drive(db, work, lowest):
    """work[t] = list of (reads, mates) batches of thread t; -> results[t][i] = candidates of that batch"""
    L = api.lib()
    K = db.cfg.max_candidates
    out = [[None] * len(w) for w in work]
    errs = []

    def run(t):
        try:
            for i, (reads, mates) in enumerate(work[t]):
                for j, r in enumerate(reads):
                    m = mates[j] if mates else b""
                    mw = db.max_windows_in_range(len(r), len(m), 0)
                    rc = L.mc_batch_add(db.h, t, r, len(r), m if mates else None, len(m), mw)
                    assert rc == 0, rc
                db._check(L.mc_batch_submit(db.h, t, lowest))
                res = api.McResults()
                db._check(L.mc_batch_wait(db.h, t, C.byref(res)))
                assert res.num_queries == len(reads)
                out[t][i] = api._view(res.cands, len(reads) * K, api.cand_dtype).reshape(len(reads), K).copy()
                db._check(L.mc_batch_clear(db.h, t))
        except Exception as e:                                     # noqa: BLE001
            errs.append(repr(e))

    th = [threading.Thread(target=run, args=(t,)) for t in range(len(work))]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errs, errs
    return out


@pytest.mark.parametrize("coalesce", ["1", "0"])
@pytest.mark.parametrize("lowest", [0, 4])
def test_slots_from_many_threads_get_their_own_candidates(golden, monkeypatch, coalesce, lowest):
    monkeypatch.setenv("MC_SLOT_COALESCE", coalesce)
    single, p1, p2 = golden.reads()
    T = 6
    db = api.Database.open(golden.db_path("toy32"), max_candidates=2, num_slots=T, slot_max_queries=96, slot_max_chars=96 * 720)
    try:
        L = api.lib()
        L.mc_batch_add.argtypes = [C.c_void_p, C.c_uint32, C.c_char_p, C.c_uint32, C.c_char_p, C.c_uint32, C.c_uint32]
        L.mc_slot_stats.argtypes = [C.c_void_p, C.c_void_p]
        rng = np.random.default_rng(77)
        work = []
        for t in range(T):
            mine = []
            for i in range(8):
                n = int(rng.integers(0 if i == 3 else 1, 90))       # (an empty slot among them)
                if (t + i) % 3 == 0:
                    idx = rng.integers(0, len(p1), n)
                    mine.append(([p1[k] for k in idx], [p2[k] for k in idx]))
                else:
                    idx = rng.integers(0, len(single), n)
                    mine.append(([single[k] for k in idx if len(single[k]) < 700], None))
            work.append(mine)
        got = _drive(db, work, lowest)
        st = (C.c_uint64 * 4)()
        assert L.mc_slot_stats(db.h, st) == 0
        assert bool(st[0]) == (coalesce == "1")                  # (slots of 96 reads: united by default as well)
        if coalesce == "1":
            assert st[2] == sum(1 for w in work for (r, _) in w if len(r)) and 1 <= st[1] <= st[2] and st[3] >= 1
        for t in range(T):
            for i, (reads, mates) in enumerate(work[t]):
                if not reads:
                    assert got[t][i].shape[0] == 0
                    continue
                want = _alone(db, reads, mates, lowest)
                assert np.array_equal(got[t][i], want), (t, i)
    finally:
        db.close()
