"""ctypes bindings for the two CPU checkers (TEST INFRASTRUCTURE):

* ``oracle()``      -> oracle/libmc_oracle.so   (our C restatement; symbols ``mco_*``)
* ``reference(b)``  -> oracle/_ref/libmcref_u{32,16}.so (the real reference + ref_shim.cpp; ``ref_*``)

Both export the same calls, so one wrapper class serves both.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

NUM_RANKS = 21
RANKS = ["sequence", "form", "variety", "subspecies", "species", "subgenus", "genus", "subtribe", "tribe",
         "subfamily", "family", "suborder", "order", "subclass", "class", "subphylum", "phylum",
         "subkingdom", "kingdom", "domain", "root"]

hit_dtype = np.dtype([("win", "<u4"), ("tgt", "<u4")])
cand_dtype = np.dtype([("taxid", "<i8"), ("tgt", "<u4"), ("hits", "<u4"), ("beg", "<u4"), ("end", "<u4")])
assert cand_dtype.itemsize == 24 and hit_dtype.itemsize == 8

_u8p = C.POINTER(C.c_uint8)


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


class CpuRef:
    def __init__(self, path: str, prefix: str):
        self.lib = C.CDLL(path)
        self.prefix = prefix
        self.is_oracle = prefix == "mco_"
        f = self._f
        f("sketch").restype = C.c_int64
        f("sketch").argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                C.c_void_p, C.c_void_p, C.c_uint64]
        f("db_open").restype = C.c_void_p
        f("db_open").argtypes = [C.c_char_p]
        f("db_close").argtypes = [C.c_void_p]
        f("db_info").argtypes = [C.c_void_p, C.c_void_p]
        f("db_max_locations_per_feature").argtypes = [C.c_void_p, C.c_uint64]
        f("db_remove_features_with_more_locations_than").restype = C.c_uint64
        f("db_remove_features_with_more_locations_than").argtypes = [C.c_void_p, C.c_uint64]
        f("db_lineages").argtypes = [C.c_void_p, C.c_void_p]
        f("db_target_name").restype = C.c_int64
        f("db_target_name").argtypes = [C.c_void_p, C.c_uint64, C.c_char_p, C.c_uint64]
        f("handler_new").restype = C.c_void_p
        f("handler_free").argtypes = [C.c_void_p]
        q = f("query")
        q.restype = C.c_int
        base = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
                C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_int, C.c_uint64]
        if self.is_oracle:
            base.append(C.c_int)
        q.argtypes = base + [C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        qm = f("query_many")
        qm.restype = C.c_double
        qm.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.c_uint64, C.c_int, C.c_void_p]
        if self.is_oracle:
            self.lib.mco_query_many_pairs.restype = C.c_double
            self.lib.mco_query_many_pairs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.c_uint64,
                                                      C.c_int, C.c_void_p]
        if self.is_oracle:
            self.lib.mco_candidates.restype = C.c_uint64
            self.lib.mco_candidates.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_void_p, C.c_int,
                                                C.c_void_p, C.c_uint64]
            self.lib.mco_db_lookup.restype = C.c_uint32
            self.lib.mco_db_lookup.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]

            self.lib.mco_db_build_claim.restype = C.c_void_p
            self.lib.mco_db_build_claim.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, C.c_void_p,
                                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int, C.c_uint32, C.c_void_p]
            self.lib.mco_db_part_arrays.restype = C.c_uint64
            self.lib.mco_db_part_arrays.argtypes = [C.c_void_p, C.c_uint32] + [C.POINTER(C.c_void_p)] * 4

    def _f(self, name):
        return getattr(self.lib, self.prefix + name)

    def build_db(self, lengths: np.ndarray, gen_addr: int, gen_user: int, wanted: np.ndarray | None = None,
                 lineage: np.ndarray | None = None, threads: int = 1, k=16, s=16, w=127, stride=112, max_locs=254,
                 target_bytes=4, claim: int = 1) -> "CpuDb":
        """The oracle's restatement of the database BUILD (mco_db_build), optionally restricted to the features in `wanted`.
        gen_addr = address of  void gen(void* user, uint32_t target, char* dst)  which writes target `target` (lengths[target]
        characters); a ctypes CFUNCTYPE object or a C function of another library (metacache_amd/synth)."""
        assert self.is_oracle
        lengths = np.ascontiguousarray(lengths, dtype=np.uint32)
        w_arr = None if wanted is None else np.ascontiguousarray(wanted, dtype=np.uint32)
        lin = None if lineage is None else np.ascontiguousarray(lineage, dtype=np.int64)
        tw = np.zeros(len(lengths), dtype=np.uint64)
        h = self.lib.mco_db_build_claim(k, s, w, stride, max_locs, target_bytes, len(lengths), _ptr(lengths), gen_addr, gen_user,
                                        None if w_arr is None else _ptr(w_arr), 0 if w_arr is None else len(w_arr),
                                        None if lin is None else _ptr(lin), threads, claim, _ptr(tw))
        if not h:
            raise RuntimeError("mco_db_build failed")
        db = CpuDb(self, h)
        db.target_windows = tw
        return db

    # ---- sketching -------------------------------------------------------------------------
    def sketch(self, seq: bytes, k=16, s=16, w=127, stride=112):
        """-> (feats[nwin, s] uint32 padded with 0xFFFFFFFF, counts[nwin])"""
        seq = bytes(seq)
        maxw = len(seq) // max(stride, 1) + 2
        feats = np.empty((maxw, s), dtype=np.uint32)
        counts = np.empty(maxw, dtype=np.uint32)
        buf = C.create_string_buffer(seq, len(seq) + 1)
        n = self._f("sketch")(C.cast(buf, C.c_void_p), len(seq), k, s, w, stride, _ptr(feats), _ptr(counts), maxw)
        assert n >= 0, n
        return feats[:n].copy(), counts[:n].copy()

    # ---- database --------------------------------------------------------------------------
    def open(self, name: str) -> "CpuDb":
        h = self._f("db_open")(name.encode())
        if not h:
            raise RuntimeError(f"cannot open database {name}")
        return CpuDb(self, h)

    def open_part(self, name: str, part: int) -> "CpuDb":
        assert self.is_oracle
        self.lib.mco_db_open_part.restype = C.c_void_p
        self.lib.mco_db_open_part.argtypes = [C.c_char_p, C.c_int]
        h = self.lib.mco_db_open_part(name.encode(), part)
        if not h:
            raise RuntimeError(f"cannot open part {part} of {name}")
        return CpuDb(self, h)

    def candidates(self, locs_u64: np.ndarray, max_win: int, max_cand: int, taxkey: np.ndarray | None = None,
                   merge: bool = False) -> np.ndarray:
        assert self.is_oracle
        locs = np.ascontiguousarray(locs_u64, dtype=np.uint64)
        cap = max(len(locs), 1)
        out = np.zeros(cap, dtype=cand_dtype)
        tk = None if taxkey is None else np.ascontiguousarray(taxkey, dtype=np.int64)
        n = self.lib.mco_candidates(_ptr(locs), len(locs), max_win, max_cand, None if tk is None else _ptr(tk),
                                    int(merge), _ptr(out), cap)
        return out[:n].copy()


class CpuDb:
    def __init__(self, ref: CpuRef, h):
        self.ref, self.h = ref, C.c_void_p(h)
        info = np.zeros(8, dtype=np.uint64)
        ref._f("db_info")(self.h, _ptr(info))
        (self.k, self.s, self.w, self.stride, self.max_locs, self.n_targets, self.n_parts, self.n_locations) = map(int, info)
        self._handler = C.c_void_p(ref._f("handler_new")())

    def close(self):
        if self.h:
            self.ref._f("handler_free")(self._handler)
            self.ref._f("db_close")(self.h)
            self.h = None

    def info(self):
        info = np.zeros(8, dtype=np.uint64)
        self.ref._f("db_info")(self.h, _ptr(info))
        return list(map(int, info))

    def set_max_locations_per_feature(self, n: int):
        self.ref._f("db_max_locations_per_feature")(self.h, n)

    def remove_features_with_more_locations_than(self, n: int) -> int:
        return int(self.ref._f("db_remove_features_with_more_locations_than")(self.h, n))

    def lineages(self) -> np.ndarray:
        out = np.zeros((self.n_targets, NUM_RANKS), dtype=np.int64)
        self.ref._f("db_lineages")(self.h, _ptr(out))
        return out

    def target_name(self, t: int) -> str:
        buf = C.create_string_buffer(4096)
        n = self.ref._f("db_target_name")(self.h, t, buf, 4096)
        return buf.raw[:n].decode() if n >= 0 else ""

    def part_arrays(self, part: int = 0):
        """-> (keys u32[n], sizes u32[n], offs u64[n], values u64[...]) views of one part (oracle only)"""
        assert self.ref.is_oracle
        pk, ps, po, pv = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        n = int(self.ref.lib.mco_db_part_arrays(self.h, part, C.byref(pk), C.byref(ps), C.byref(po), C.byref(pv)))
        if n == 0:
            return (np.zeros(0, np.uint32), np.zeros(0, np.uint32), np.zeros(0, np.uint64), np.zeros(0, np.uint64))
        keys = np.ctypeslib.as_array(C.cast(pk, C.POINTER(C.c_uint32)), shape=(n,))
        sizes = np.ctypeslib.as_array(C.cast(ps, C.POINTER(C.c_uint32)), shape=(n,))
        offs = np.ctypeslib.as_array(C.cast(po, C.POINTER(C.c_uint64)), shape=(n,))
        nv = int(offs[-1]) + 255
        vals = np.ctypeslib.as_array(C.cast(pv, C.POINTER(C.c_uint64)), shape=(nv,))
        return keys, sizes, offs, vals

    def lookup(self, feature: int, part: int = 0) -> np.ndarray:
        assert self.ref.is_oracle
        p = C.c_void_p()
        n = self.ref.lib.mco_db_lookup(self.h, part, feature, C.byref(p))
        if n == 0:
            return np.zeros(0, dtype=np.uint64)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint64)), shape=(n,)).copy()

    def new_handler(self):
        """a query handler of its own for a caller thread (query(..., handler=h); the default one belongs to the CpuDb: one thread at a time)"""
        return C.c_void_p(self.ref._f("handler_new")())

    def free_handler(self, h):
        self.ref._f("handler_free")(h)

    def query(self, s1: bytes, s2: bytes = b"", max_cand: int = 2, lowest: int = 0, insert_max: int = 0,
              sketchlen: int = 0, winlen: int = 0, winstride: int = 0, mode: int = 0, handler=None):
        """-> (allhits[hit_dtype], tophits[cand_dtype])"""
        s1, s2 = bytes(s1), bytes(s2)
        b1 = C.create_string_buffer(s1, len(s1) + 1)
        b2 = C.create_string_buffer(s2, len(s2) + 1)
        pa, na, pt, nt = C.c_void_p(), C.c_uint64(), C.c_void_p(), C.c_uint64()
        args = [self.h, handler if handler is not None else self._handler, C.cast(b1, C.c_void_p), len(s1), C.cast(b2, C.c_void_p), len(s2),
                sketchlen, winlen, winstride, max_cand, lowest, insert_max]
        if self.ref.is_oracle:
            args.append(mode)
        self.ref._f("query")(*args, C.byref(pa), C.byref(na), C.byref(pt), C.byref(nt))
        hits = np.zeros(na.value, dtype=hit_dtype)
        if na.value:
            C.memmove(_ptr(hits), pa, na.value * 8)
        cands = np.zeros(nt.value, dtype=cand_dtype)
        if nt.value:
            C.memmove(_ptr(cands), pt, nt.value * 24)
        return hits, cands

    def query_many(self, seqs: np.ndarray, offs: np.ndarray, max_cand: int = 2, lowest: int = 0, insert_max: int = 0,
                   threads: int = 1, want_cands: bool = True):
        """-> (seconds, cands[n, max_cand] or None)"""
        seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        n = len(offs) - 1
        cands = np.zeros((n, max_cand), dtype=cand_dtype) if want_cands else None
        t = self.ref._f("query_many")(self.h, _ptr(seqs), _ptr(offs), n, max_cand, lowest, insert_max, threads,
                                      None if cands is None else _ptr(cands))
        return t, cands


def _query_many_pairs(self, seqs, offs, seqs2, offs2, max_cand: int = 2, lowest: int = 0, insert_max: int = 0, threads: int = 1):
    """read pairs through the oracle's threaded bulk entry (mco_query_many_pairs) -> (seconds, cands[n, max_cand])"""
    seqs = np.ascontiguousarray(seqs, dtype=np.uint8); seqs2 = np.ascontiguousarray(seqs2, dtype=np.uint8)
    offs = np.ascontiguousarray(offs, dtype=np.uint64); offs2 = np.ascontiguousarray(offs2, dtype=np.uint64)
    n = len(offs) - 1
    cands = np.zeros((n, max_cand), dtype=cand_dtype)
    t = self.ref.lib.mco_query_many_pairs(self.h, _ptr(seqs), _ptr(offs), _ptr(seqs2), _ptr(offs2), n, max_cand, lowest, insert_max, threads, _ptr(cands))
    return t, cands


CpuDb.query_many_pairs = _query_many_pairs


def build_oracle() -> str:
    so = os.path.join(ORACLE_DIR, "libmc_oracle.so")
    src = os.path.join(ORACLE_DIR, "mc_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "oracle"], stdout=subprocess.DEVNULL)
    return so


_cache: dict = {}


def oracle() -> CpuRef:
    if "o" not in _cache:
        _cache["o"] = CpuRef(build_oracle(), "mco_")
    return _cache["o"]


def reference_path(target_bytes: int = 4) -> str:
    return os.path.join(ORACLE_DIR, "_ref", f"libmcref_u{target_bytes * 8}.so")


def have_reference(target_bytes: int = 4) -> bool:
    return os.path.exists(reference_path(target_bytes))


def reference(target_bytes: int = 4) -> CpuRef:
    key = ("r", target_bytes)
    if key not in _cache:
        _cache[key] = CpuRef(reference_path(target_bytes), "ref_")
    return _cache[key]
