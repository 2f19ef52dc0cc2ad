#!/usr/bin/env python3
"""toy32p4.meta / .cache{0..3}: a FOUR-part database for the part-group tests (mc_partset_*).

The reference's own `build -parts 4` cannot write it from 24 small genomes (a part stays empty and its writer divides by zero, SURVEY
8c), so the reference-written TWO-part fixture (toy32p2, make_golden.py) is re-cut: part p of toy32p2 becomes the parts 2p (its targets
with an even id) and 2p + 1 (odd id) -- every bucket's locations dealt out by target, order kept (buckets are sorted by (target,
window), hash_multimap.hpp:1037-1082 batch layout), empty buckets dropped, the part count in the .meta header set to 4.  What the
four parts hold together is exactly what the two held; the expected classifications come from the oracle's intended multi-part
semantics on the re-cut files (tests/test_gpu_parts.py), the same oracle that is pinned on the reference for toy32p2.
Data transformation of a committed fixture: python tests/golden/make_golden_parts.py"""
import os
import struct

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
BATCH = 1 << 20


def read_part(fn):
    b = open(fn, "rb").read()
    nkeys, nvalues, batch = struct.unpack_from("<QQQ", b, 0)
    off = 24
    keys, sizes, vals = [], [], []
    done = 0
    while done < nkeys:
        nb = min(batch, nkeys - done)
        k = np.frombuffer(b, dtype="<u4", count=nb, offset=off); off += 4 * nb
        s = np.frombuffer(b, dtype="u1", count=nb, offset=off); off += nb
        nv = int(s.sum())
        v = np.frombuffer(b, dtype="<u4", count=2 * nv, offset=off).reshape(nv, 2); off += 8 * nv      # {win, tgt}
        keys.append(k); sizes.append(s); vals.append(v)
        done += nb
    assert off == len(b)
    return np.concatenate(keys), np.concatenate(sizes), np.concatenate(vals)


def write_part(fn, keys, sizes, vals):
    with open(fn, "wb") as f:
        f.write(struct.pack("<QQQ", len(keys), len(vals), BATCH))
        vo = np.zeros(len(keys) + 1, dtype=np.int64)
        vo[1:] = np.cumsum(sizes)
        for a in range(0, len(keys), BATCH):
            e = min(len(keys), a + BATCH)
            f.write(keys[a:e].astype("<u4").tobytes()); f.write(sizes[a:e].astype("u1").tobytes())
            f.write(vals[vo[a]:vo[e]].astype("<u4").tobytes())


def main():
    meta = bytearray(open(os.path.join(HERE, "toy32p2.meta"), "rb").read())
    assert struct.unpack_from("<I", meta, 91)[0] == 2          # u64 version, 7 x u8, 2 x 4 x u64 sketching, u64 max locations, u32 targets, u32 PARTS
    struct.pack_into("<I", meta, 91, 4)
    open(os.path.join(HERE, "toy32p4.meta"), "wb").write(bytes(meta))
    for p in range(2):
        keys, sizes, vals = read_part(os.path.join(HERE, f"toy32p2.cache{p}"))
        key_of = np.repeat(np.arange(len(keys)), sizes)
        for odd in range(2):
            sel = (vals[:, 1] & 1) == odd
            cnt = np.bincount(key_of[sel], minlength=len(keys))
            keep = cnt > 0
            write_part(os.path.join(HERE, f"toy32p4.cache{2 * p + odd}"), keys[keep], cnt[keep], vals[sel])
            print(f"toy32p4.cache{2 * p + odd}: {int(keep.sum())} keys, {int(sel.sum())} locations")


if __name__ == "__main__":
    main()
