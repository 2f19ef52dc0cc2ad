"""ctypes binding of the C ABI in include/metacache_amd.h (the product's only entry points).

There is deliberately no CPU path here: if libmetacache_amd.so cannot be built / loaded, or no GPU
is usable, every call raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import build as _build

MC_OK, MC_BATCH_FULL = 0, 1
NUM_RANKS = 21

cand_dtype = np.dtype([("tgt", "<u4"), ("hits", "<u4"), ("beg", "<u4"), ("end", "<u4")])
loc_dtype = np.dtype([("win", "<u4"), ("tgt", "<u4")])
qstat_dtype = np.dtype([("hits", "<u4"), ("nfeat", "<u4"), ("nfound", "<u4"), ("nsteps", "<u4")])


class McConfig(C.Structure):
    _fields_ = [("device", C.c_int32), ("kmerlen", C.c_uint32), ("sketchlen", C.c_uint32), ("winlen", C.c_uint32),
                ("winstride", C.c_uint32), ("max_candidates", C.c_uint32), ("target_id_bytes", C.c_uint32),
                ("num_parts", C.c_uint32), ("max_locations_per_feature", C.c_uint32), ("remove_overpopulated", C.c_uint32),
                ("max_load_factor", C.c_float), ("num_slots", C.c_uint32), ("slot_max_queries", C.c_uint32),
                ("slot_max_chars", C.c_uint32), ("copy_allhits", C.c_uint32), ("single_part", C.c_int32),
                ("key_shard_index", C.c_uint32), ("key_shard_count", C.c_uint32),
                ("target_shard_index", C.c_uint32), ("target_shard_count", C.c_uint32)]


class McResults(C.Structure):
    _fields_ = [("num_queries", C.c_uint32), ("max_candidates", C.c_uint32), ("cands", C.c_void_p),
                ("hit_offsets", C.c_void_p), ("hits", C.c_void_p), ("hit_counts", C.c_void_p)]


class McDeviceBatch(C.Structure):
    _fields_ = [("seq", C.c_void_p), ("qinfo", C.c_void_p), ("max_win", C.c_void_p), ("max_win_uniform", C.c_uint32),
                ("num_queries", C.c_uint32), ("num_chars", C.c_uint64)]


class McDeviceHits(C.Structure):
    _fields_ = [("hits", C.c_void_p), ("hit_offsets", C.c_void_p), ("max_win", C.c_void_p), ("max_win_uniform", C.c_uint32),
                ("num_queries", C.c_uint32)]


class McDevicePartialHits(C.Structure):
    _fields_ = [("counts", C.c_void_p), ("hits", C.c_void_p), ("total_hits", C.c_uint64), ("max_win", C.c_void_p), ("max_win_uniform", C.c_uint32),
                ("num_queries", C.c_uint32), ("num_sources", C.c_uint32)]


class McDevicePartialNumbers(C.Structure):
    _fields_ = [("counts", C.c_void_p), ("numbers", C.c_void_p), ("total", C.c_uint64)]


class McDevicePartialNumbersIn(C.Structure):
    _fields_ = [("counts", C.c_void_p), ("numbers", C.c_void_p), ("source_offsets", C.c_void_p), ("max_win", C.c_void_p), ("max_win_uniform", C.c_uint32),
                ("num_queries", C.c_uint32), ("num_sources", C.c_uint32)]


class McDeviceResults(C.Structure):
    _fields_ = [("cands", C.c_void_p), ("hit_counts", C.c_void_p), ("hit_offsets", C.c_void_p), ("hits", C.c_void_p),
                ("features", C.c_void_p), ("win_offsets", C.c_void_p)]


EXPORTS = ["mc_candidates_from_partial_numbers_on", "mc_runtime_warning", "mc_slot_stats", "mc_config_default", "mc_create", "mc_destroy", "mc_last_error", "mc_load_begin", "mc_load_batch", "mc_load_end", "mc_load_location_range", "mc_load_target_windows", "mc_table_layout", "mc_target_range", "mc_merge_part_candidates", "mc_partset_open", "mc_partset_close",
           "mc_partset_info", "mc_partset_classify", "mc_partset_last_error", "mc_partset_select_group", "mc_partset_classify_resident", "mc_partset_load_bytes",
           "mc_partial_numbers", "mc_candidates_from_partial_numbers", "mc_owner_stats", "mc_keyset_open", "mc_keyset_close", "mc_keyset_info", "mc_keyset_classify", "mc_keyset_last_error",
           "mc_open_database", "mc_open_metadata", "mc_load_stats", "mc_set_lineages", "mc_db_info", "mc_db_num_taxa", "mc_db_taxon", "mc_db_taxon_source", "mc_db_lineages",
           "mc_batch_add", "mc_batch_add_bulk", "mc_batch_submit", "mc_batch_wait", "mc_batch_clear", "mc_query_device", "mc_query_finish", "mc_query_wait", "mc_synchronize",
           "mc_key_owner", "mc_candidates_from_hits", "mc_candidates_from_partial_hits", "mc_copy_results",
           "mc_timing_enable", "mc_timing_reset", "mc_timing_get", "mc_last_batch_stats", "mc_set_tuning", "mc_copy_results_on",
           "mc_build_begin", "mc_build_add_target", "mc_build_add_target_src", "mc_build_add_target_device", "mc_build_flush", "mc_build_reserve",
           "mc_build_table_begin", "mc_build_table_add", "mc_build_table_end", "mc_build_set_parent", "mc_build_target_windows", "mc_build_remove_ambiguous", "mc_build_counts", "mc_build_add_existing_target", "mc_build_add_locations", "mc_build_finish", "mc_build_finish_shards", "mc_build_write_shards", "mc_build_write", "mc_build_write_begin", "mc_build_write_add", "mc_build_write_end", "mc_build_free", "mc_build_last_error",
           "mc_build_set_query_config"]

_lib = None


def lib() -> C.CDLL:
    """Loads (building first if needed) libmetacache_amd.so.  Raises if that is impossible."""
    global _lib
    if _lib is None:
        # PyTorch's wheel bundles its own libamdhip64 / libhsa-runtime64.  Two HSA runtimes in one
        # process do not both see the GPU, so when torch is present it is imported FIRST and our
        # library then binds (by SONAME libamdhip64.so.7) to the runtime that is already loaded.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        path = os.environ.get("MC_AMD_LIB") or _build.build_library()   # (MC_AMD_LIB: another build of the library, for A/B measurements)
        L = C.CDLL(path)
        L.mc_last_error.restype = C.c_char_p
        L.mc_last_error.argtypes = [C.c_void_p]
        L.mc_create.argtypes = [C.POINTER(McConfig), C.POINTER(C.c_void_p)]
        L.mc_destroy.argtypes = [C.c_void_p]
        L.mc_destroy.restype = None
        L.mc_open_database.argtypes = [C.c_char_p, C.POINTER(McConfig), C.POINTER(C.c_void_p)]
        L.mc_load_begin.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint64]
        L.mc_load_batch.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
        L.mc_load_end.argtypes = [C.c_void_p, C.c_uint32]
        L.mc_load_location_range.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        L.mc_load_target_windows.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        L.mc_table_layout.argtypes = [C.c_void_p, C.c_void_p]
        L.mc_target_range.argtypes = [C.c_void_p, C.c_void_p]
        L.mc_set_lineages.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        L.mc_db_info.argtypes = [C.c_void_p, C.c_void_p]
        L.mc_db_num_taxa.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L.mc_db_taxon.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_uint32),
                                  C.POINTER(C.c_char_p)]
        L.mc_db_lineages.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        L.mc_batch_add.argtypes = [C.c_void_p, C.c_uint32, C.c_char_p, C.c_uint32, C.c_char_p, C.c_uint32, C.c_uint32]
        L.mc_batch_add_bulk.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64]
        L.mc_batch_add_bulk.restype = C.c_int64
        L.mc_batch_submit.argtypes = [C.c_void_p, C.c_uint32, C.c_int]
        L.mc_batch_wait.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(McResults)]
        L.mc_batch_clear.argtypes = [C.c_void_p, C.c_uint32]
        L.mc_query_device.argtypes = [C.c_void_p, C.POINTER(McDeviceBatch), C.c_int, C.c_int, C.POINTER(McDeviceResults), C.c_void_p]
        L.mc_query_finish.argtypes = [C.c_void_p, C.c_int]
        L.mc_query_wait.argtypes = [C.c_void_p, C.c_int]
        L.mc_synchronize.argtypes = [C.c_void_p]
        L.mc_key_owner.argtypes = [C.c_uint32, C.c_uint32]
        L.mc_key_owner.restype = C.c_uint32
        L.mc_candidates_from_hits.argtypes = [C.c_void_p, C.POINTER(McDeviceHits), C.c_int, C.POINTER(McDeviceResults), C.c_void_p]
        L.mc_copy_results.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int]
        L.mc_timing_enable.argtypes = [C.c_void_p, C.c_int]
        L.mc_timing_reset.argtypes = [C.c_void_p]
        L.mc_timing_get.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
        L.mc_last_batch_stats.argtypes = [C.c_void_p, C.c_void_p]
        if hasattr(L, "mc_build_begin"):
            L.mc_build_begin.argtypes = [C.POINTER(McConfig), C.POINTER(C.c_void_p)]
            L.mc_build_add_target.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_char_p, C.c_int64, C.c_char_p]
            L.mc_build_finish.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
            L.mc_build_write.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint64]
            L.mc_build_free.argtypes = [C.c_void_p]
            L.mc_build_free.restype = None
        _lib = L
    return _lib


class McError(RuntimeError):
    pass


def default_config(**kw) -> McConfig:
    cfg = McConfig()
    lib().mc_config_default(C.byref(cfg))
    for k, v in kw.items():
        setattr(cfg, k, v)
    return cfg


def _view(ptr, n, dtype):
    if not ptr or n == 0:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_uint8 * (n * dtype.itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=n)


class Database:
    """Host-side mirror of what the reference's `database` gives the query layer
    (database.hpp:386-408): open files, run batches, read results."""

    def __init__(self, handle, cfg: McConfig):
        self.h = C.c_void_p(handle)
        self.cfg = cfg
        self.refresh_info()

    def refresh_info(self):
        info = np.zeros(8, dtype=np.uint64)
        self._check(lib().mc_db_info(self.h, info.ctypes.data_as(C.c_void_p)))
        (self.k, self.s, self.w, self.stride, self.max_locs, self.n_targets, self.n_parts, self.n_locations) = map(int, info)

    # ---- construction ----------------------------------------------------------------------
    @classmethod
    def open(cls, name: str, **kw) -> "Database":
        kw.setdefault("kmerlen", 0); kw.setdefault("sketchlen", 0); kw.setdefault("winlen", 0); kw.setdefault("winstride", 0)
        cfg = default_config(**kw)
        h = C.c_void_p()
        rc = lib().mc_open_database(name.encode(), C.byref(cfg), C.byref(h))
        if rc != MC_OK:
            raise McError(f"mc_open_database({name}) -> {rc}: {lib().mc_last_error(None).decode()}")
        return cls(h.value, cfg)

    @classmethod
    def from_handle(cls, handle, cfg) -> "Database":
        return cls(handle, cfg)

    def close(self):
        if self.h:
            lib().mc_destroy(self.h)
            self.h = None

    def _check(self, rc):
        if rc < 0:
            raise McError(f"metacache_amd error {rc}: {lib().mc_last_error(self.h).decode()}")
        return rc

    def info(self):
        info = np.zeros(8, dtype=np.uint64)
        self._check(lib().mc_db_info(self.h, info.ctypes.data_as(C.c_void_p)))
        return list(map(int, info))

    def set_lineages(self, lin: np.ndarray):
        """lin[targets, 21] uint32: taxon index + 1 per rank, 0 = none (mc_set_lineages) -- needed for lowest > 0 on contexts that
        were not opened from database files"""
        lin = np.ascontiguousarray(lin, dtype=np.uint32)
        assert lin.ndim == 2 and lin.shape[1] == 21
        self._check(lib().mc_set_lineages(self.h, lin.ctypes.data_as(C.c_void_p), lin.shape[0]))

    # ---- taxonomy --------------------------------------------------------------------------
    def taxa(self):
        n = C.c_uint64()
        self._check(lib().mc_db_num_taxa(self.h, C.byref(n)))
        out = []
        for i in range(n.value):
            tid, par, rk, nm = C.c_int64(), C.c_int64(), C.c_uint32(), C.c_char_p()
            self._check(lib().mc_db_taxon(self.h, i, C.byref(tid), C.byref(par), C.byref(rk), C.byref(nm)))
            out.append((tid.value, par.value, rk.value, nm.value.decode()))
        return out

    def lineages(self) -> np.ndarray:
        p, n = C.c_void_p(), C.c_uint64()
        self._check(lib().mc_db_lineages(self.h, C.byref(p), C.byref(n)))
        return _view(p.value, n.value * NUM_RANKS, np.dtype("<u4")).reshape(n.value, NUM_RANKS).copy()

    def max_windows_in_range(self, l1: int, l2: int = 0, insert_max: int = 0) -> int:
        """candidate_structs.hpp:143-145 (stride = the database's window stride)"""
        return (2 + max(l1 + l2, insert_max) // self.stride) & 0xFFFFFFFF

    # ---- host slot path ---------------------------------------------------------------------
    def query(self, reads, mates=None, lowest: int = 0, insert_max: int = 0, slot: int = 0):
        """Runs all reads (bytes) through slot batches.
        -> (cands[n, K] cand_dtype, hit_counts[n], allhits list or None)"""
        L = lib()
        n = len(reads)
        K = self.cfg.max_candidates
        cands = np.zeros((n, K), dtype=cand_dtype)
        counts = np.zeros(n, dtype=np.uint32)
        allhits = [None] * n if self.cfg.copy_allhits else None
        start = 0

        def flush(upto):
            nonlocal start
            self._check(L.mc_batch_submit(self.h, slot, lowest))
            res = McResults()
            self._check(L.mc_batch_wait(self.h, slot, C.byref(res)))
            m = res.num_queries
            assert m == upto - start
            cands[start:upto] = _view(res.cands, m * K, cand_dtype).reshape(m, K)
            counts[start:upto] = _view(res.hit_counts, m, np.dtype("<u4"))
            if allhits is not None:
                off = _view(res.hit_offsets, m + 1, np.dtype("<u8"))
                hits = _view(res.hits, int(off[m]), loc_dtype)
                for i in range(m):
                    allhits[start + i] = hits[int(off[i]):int(off[i + 1])].copy()
            self._check(L.mc_batch_clear(self.h, slot))
            start = upto

        for i in range(n):
            a = bytes(reads[i]); b = bytes(mates[i]) if mates is not None else b""
            mw = self.max_windows_in_range(len(a), len(b), insert_max)
            rc = self._check(L.mc_batch_add(self.h, slot, a, len(a), b, len(b), mw))
            if rc == MC_BATCH_FULL:
                flush(i)
                rc = self._check(L.mc_batch_add(self.h, slot, a, len(a), b, len(b), mw))
                if rc != MC_OK:
                    raise McError("query does not fit an empty slot")
        flush(n)
        return cands, counts, allhits

    def query_bulk(self, seqs: np.ndarray, offs: np.ndarray, lowest: int = 0, insert_max: int = 0, slot: int = 0) -> np.ndarray:
        """single-end reads given as one byte array + offsets; host slot path (H2D of the characters, D2H of the
        candidates) -> cands[n, K]"""
        L = lib()
        seqs = np.ascontiguousarray(seqs, dtype=np.uint8); offs = np.ascontiguousarray(offs, dtype=np.uint64)
        n = len(offs) - 1
        K = self.cfg.max_candidates
        out = np.zeros((n, K), dtype=cand_dtype)
        done = 0
        while done < n:
            added = L.mc_batch_add_bulk(self.h, slot, seqs.ctypes.data_as(C.c_void_p), offs[done:].ctypes.data_as(C.c_void_p), n - done, insert_max)
            self._check(added)
            if added == 0:
                raise McError("query does not fit an empty slot")
            self._check(L.mc_batch_submit(self.h, slot, lowest))
            res = McResults()
            self._check(L.mc_batch_wait(self.h, slot, C.byref(res)))
            out[done:done + added] = _view(res.cands, added * K, cand_dtype).reshape(added, K)
            self._check(L.mc_batch_clear(self.h, slot))
            done += added
        return out

    # ---- device path (pointers are device addresses, e.g. torch tensors' data_ptr()) ----------
    def query_device(self, seq_ptr: int, qinfo_ptr: int, n: int, num_chars: int, max_win_ptr: int = 0, max_win_uniform: int = 0,
                     lowest: int = 0, want_allhits: bool = False, want_features: bool = False,
                     stream: int = 0, want_partial_hits: bool = False, second_pipe: bool = False, want_partial_numbers: bool = False,
                     defer_tail: bool = False) -> McDeviceResults:
        b = McDeviceBatch(seq_ptr, qinfo_ptr, max_win_ptr or None, max_win_uniform, n, num_chars)
        r = McDeviceResults()
        self._check(lib().mc_query_device(self.h, C.byref(b), lowest, int(want_allhits) | (2 if want_features else 0) | (4 if want_partial_hits else 0) | (8 if second_pipe else 0) | (16 if want_partial_numbers else 0) | (32 if defer_tail else 0), C.byref(r),
                                          stream or None))
        return r

    def query_finish(self, second_pipe: bool = False):
        """the tail of the last query_device(defer_tail=True) call on that pipe (mc_query_finish)"""
        self._check(lib().mc_query_finish(self.h, 8 if second_pipe else 0))

    def query_wait(self, second_pipe: bool = False):
        """waits for ONE pipe's stream (mc_query_wait)"""
        self._check(lib().mc_query_wait(self.h, 8 if second_pipe else 0))

    def candidates_from_hits(self, hits_ptr: int, hit_offsets_ptr: int, n: int, max_win_ptr: int = 0, max_win_uniform: int = 0,
                             lowest: int = 0, stream: int = 0) -> McDeviceResults:
        """Mode K: rows 8-10 on gathered location lists (device pointers)"""
        h = McDeviceHits(hits_ptr, hit_offsets_ptr, max_win_ptr or None, max_win_uniform, n)
        r = McDeviceResults()
        self._check(lib().mc_candidates_from_hits(self.h, C.byref(h), lowest, C.byref(r), stream or None))
        return r

    def candidates_from_partial_hits(self, counts_ptr: int, hits_ptr: int, total_hits: int, n: int, sources: int, max_win_ptr: int = 0,
                                     max_win_uniform: int = 0, lowest: int = 0, stream: int = 0) -> McDeviceResults:
        """Mode K owner side: union of the sources' partial lists + rows 8-10, all on the device (mc_candidates_from_partial_hits)"""
        L = lib()
        L.mc_candidates_from_partial_hits.argtypes = [C.c_void_p, C.POINTER(McDevicePartialHits), C.c_int, C.POINTER(McDeviceResults), C.c_void_p]
        h = McDevicePartialHits(counts_ptr, hits_ptr or None, total_hits, max_win_ptr or None, max_win_uniform, n, sources)
        r = McDeviceResults()
        self._check(L.mc_candidates_from_partial_hits(self.h, C.byref(h), lowest, C.byref(r), stream or None))
        return r

    def partial_numbers(self, res: McDeviceResults, n: int, cut_queries, stream: int = 0):
        """Mode K shard side, 4-byte wire: the partial lists of the last query_device(want_partial_hits=True) as global window numbers
        (mc_partial_numbers).  -> (McDevicePartialNumbers, cut_offsets uint64 [len(cut_queries)])"""
        L = lib()
        L.mc_partial_numbers.argtypes = [C.c_void_p, C.POINTER(McDeviceResults), C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(McDevicePartialNumbers), C.c_void_p]
        cq = np.ascontiguousarray(cut_queries, dtype=np.uint32)
        co = np.zeros(len(cq), dtype=np.uint64)
        out = McDevicePartialNumbers()
        self._check(L.mc_partial_numbers(self.h, C.byref(res), n, cq.ctypes.data, len(cq), co.ctypes.data, C.byref(out), stream or None))
        return out, co

    def candidates_from_partial_numbers(self, counts_ptr: int, numbers_ptr: int, source_offsets, n: int, max_win_ptr: int = 0,
                                        max_win_uniform: int = 0, lowest: int = 0, stream: int = 0) -> McDeviceResults:
        """Mode K owner side, 4-byte wire (mc_candidates_from_partial_numbers); source_offsets: host array [sources + 1]"""
        L = lib()
        L.mc_candidates_from_partial_numbers.argtypes = [C.c_void_p, C.POINTER(McDevicePartialNumbersIn), C.c_int, C.POINTER(McDeviceResults), C.c_void_p]
        so = np.ascontiguousarray(source_offsets, dtype=np.uint64)
        h = McDevicePartialNumbersIn(counts_ptr, numbers_ptr or None, so.ctypes.data, max_win_ptr or None, max_win_uniform, n, len(so) - 1)
        r = McDeviceResults()
        self._check(L.mc_candidates_from_partial_numbers(self.h, C.byref(h), lowest, C.byref(r), stream or None))
        return r

    def copy_results(self, dst_ptr: int, src_ptr: int, nbytes: int, to_host: bool = False, stream: int = 0, second_pipe: bool = False, from_host: bool = False):
        L = lib()
        L.mc_copy_results_on.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p]
        self._check(L.mc_copy_results_on(self.h, dst_ptr, src_ptr, nbytes, (1 if to_host else 0) | (2 if from_host else 0) | (8 if second_pipe else 0), stream or None))

    def load_stats(self) -> dict:
        """how mc_open_database read the database files (mc_load_stats)"""
        L = lib()
        L.mc_load_stats.argtypes = [C.c_void_p, C.c_void_p]
        a = (C.c_uint64 * 4)()
        self._check(L.mc_load_stats(self.h, a))
        return dict(bytes=int(a[0]), seconds=a[1] / 1e9, index_seconds=a[2] / 1e9, feeder_wait_seconds=a[3] / 1e9,
                    GB_per_s=(a[0] / a[1]) if a[1] else 0.0)

    def table_layout(self) -> dict:
        """bytes per stored location (4 = compact store: global window numbers), gap between two targets' numbers, buckets, stored list locations (mc_table_layout)"""
        a = (C.c_uint64 * 4)()
        self._check(lib().mc_table_layout(self.h, a))
        return {"location_bytes": int(a[0]) & 0xFF, "direct_index": bool(int(a[0]) >> 32), "window_gap": int(a[1]) & 0xFFFFFFFF, "list_align": int(a[1]) >> 32, "buckets": int(a[2]), "list_locations": int(a[3])}

    def target_range(self) -> tuple:
        """[lo, hi): the targets whose locations this context holds (mc_target_range)"""
        a = (C.c_uint64 * 4)()
        self._check(lib().mc_target_range(self.h, a))
        self.n_features = int(a[2])                               # (features the table holds)
        return int(a[0]), int(a[1])

    def set_tuning(self, name: str, value: int):
        L = lib()
        L.mc_set_tuning.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
        self._check(L.mc_set_tuning(self.h, name.encode(), value))

    def synchronize(self):
        self._check(lib().mc_synchronize(self.h))

    def timing(self, on: bool):
        self._check(lib().mc_timing_enable(self.h, int(on)))

    def timing_reset(self):
        self._check(lib().mc_timing_reset(self.h))

    def timing_get(self, kernel: str):
        ms, cnt = C.c_double(), C.c_uint64()
        self._check(lib().mc_timing_get(self.h, kernel.encode(), C.byref(ms), C.byref(cnt)))
        return ms.value, cnt.value

    def last_batch_stats(self):
        st = np.zeros(8, dtype=np.uint64)
        self._check(lib().mc_last_batch_stats(self.h, st.ctypes.data_as(C.c_void_p)))
        return dict(windows=int(st[0]), features=int(st[1]), locations=int(st[2]), found=int(st[3]), probe_steps=int(st[4]),
                    filtered_kept=int(st[5]), filtered_reads=int(st[6]) & 0xFFFFFFFF, filtered_over_512=int(st[6]) >> 32,
                    filter_second_kernel=int(st[7]) & 0xFFFFFFFF, filter_handed_back=int(st[7]) >> 32)


class PartSet:
    """A partitioned database, `resident` parts in HBM at a time (mc_partset_*): the next group of parts is loaded while the reads run
    against this one, per-part candidates gathered over RCCL and merged on the device in part order."""

    def __init__(self, name: str, resident: int = 0, devices=None, **kw):
        L = lib()
        L.mc_partset_open.argtypes = [C.c_char_p, C.POINTER(McConfig), C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)]
        L.mc_partset_close.argtypes = [C.c_void_p]
        L.mc_partset_info.argtypes = [C.c_void_p, C.c_void_p]
        L.mc_partset_classify.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_uint64, C.c_void_p]
        L.mc_partset_last_error.argtypes = [C.c_void_p]
        L.mc_partset_last_error.restype = C.c_char_p
        kw.setdefault("kmerlen", 0); kw.setdefault("sketchlen", 0); kw.setdefault("winlen", 0); kw.setdefault("winstride", 0)   # (the database's own, as Database.open)
        self.cfg = default_config(**kw)
        self.h = C.c_void_p()
        dv = np.asarray(devices if devices is not None else [], dtype=np.int32)
        rc = L.mc_partset_open(name.encode(), C.byref(self.cfg), resident, dv.ctypes.data if len(dv) else None, len(dv), C.byref(self.h))
        if rc != 0:
            raise McError(f"mc_partset_open({name}): {L.mc_partset_last_error(None).decode()} (rc {rc})")

    def info(self) -> dict:
        a = (C.c_uint64 * 6)()
        lib().mc_partset_info(self.h, a)
        nb = C.c_uint64()
        lib().mc_partset_load_bytes(self.h, C.byref(nb))
        return dict(parts=int(a[0]), resident=int(a[1]), groups=int(a[2]), devices=int(a[3]), load_s=a[4] / 1e9, wait_s=a[5] / 1e9, load_bytes=int(nb.value))

    def select_group(self, g: int):
        L = lib()
        L.mc_partset_select_group.argtypes = [C.c_void_p, C.c_uint32]
        rc = L.mc_partset_select_group(self.h, g)
        if rc != 0:
            raise McError(f"mc_partset_select_group: {L.mc_partset_last_error(self.h).decode()} (rc {rc})")

    def classify_resident(self, reads, mates, out: np.ndarray, has_prior: bool, lowest: int = 0, insert_max: int = 0):
        """one batch through the resident group's parts; out (cand_dtype [n, K]) holds the earlier groups' lists (has_prior) and receives the merged ones"""
        def pack(rs):
            offs = np.zeros(len(rs) + 1, dtype=np.uint64)
            offs[1:] = np.cumsum([len(r) for r in rs])
            return np.frombuffer(b"".join(rs) + b"\0", dtype=np.uint8), offs
        s1, o1 = pack(reads)
        s2, o2 = pack(mates) if mates is not None else (None, None)
        L = lib()
        L.mc_partset_classify_resident.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_uint64, C.c_int, C.c_void_p]
        rc = L.mc_partset_classify_resident(self.h, s1.ctypes.data, o1.ctypes.data, s2.ctypes.data if s2 is not None else None,
                                            o2.ctypes.data if o2 is not None else None, len(reads), lowest, insert_max, int(has_prior), out.ctypes.data)
        if rc != 0:
            raise McError(f"mc_partset_classify_resident: {L.mc_partset_last_error(self.h).decode()} (rc {rc})")

    def classify_resident_packed(self, seqs: np.ndarray, offs: np.ndarray, out: np.ndarray, has_prior: bool, lowest: int = 0):
        """classify_resident for single-end reads given as one byte array + n + 1 offsets"""
        seqs = np.ascontiguousarray(seqs, dtype=np.uint8); offs = np.ascontiguousarray(offs, dtype=np.uint64)
        L = lib()
        L.mc_partset_classify_resident.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_uint64, C.c_int, C.c_void_p]
        rc = L.mc_partset_classify_resident(self.h, seqs.ctypes.data, offs.ctypes.data, None, None, len(offs) - 1, lowest, 0, int(has_prior), out.ctypes.data)
        if rc != 0:
            raise McError(f"mc_partset_classify_resident: {L.mc_partset_last_error(self.h).decode()} (rc {rc})")

    def classify(self, reads, mates=None, lowest: int = 0, insert_max: int = 0) -> np.ndarray:
        """reads / mates: lists of bytes -> cand_dtype [n, max_candidates]"""
        def pack(rs):
            offs = np.zeros(len(rs) + 1, dtype=np.uint64)
            offs[1:] = np.cumsum([len(r) for r in rs])
            return np.frombuffer(b"".join(rs) + b"\0", dtype=np.uint8), offs
        n = len(reads)
        s1, o1 = pack(reads)
        s2, o2 = pack(mates) if mates is not None else (None, None)
        out = np.zeros((n, self.cfg.max_candidates), dtype=cand_dtype)
        L = lib()
        rc = L.mc_partset_classify(self.h, s1.ctypes.data, o1.ctypes.data, s2.ctypes.data if s2 is not None else None,
                                   o2.ctypes.data if o2 is not None else None, n, lowest, insert_max, out.ctypes.data)
        if rc != 0:
            raise McError(f"mc_partset_classify: {L.mc_partset_last_error(self.h).decode()} (rc {rc})")
        return out

    def classify_packed(self, seqs: np.ndarray, offs: np.ndarray, lowest: int = 0, insert_max: int = 0) -> np.ndarray:
        """single-end reads as one byte array + n + 1 offsets (what mc_partset_classify takes) -> cand_dtype [n, max_candidates]"""
        seqs = np.ascontiguousarray(seqs, dtype=np.uint8); offs = np.ascontiguousarray(offs, dtype=np.uint64)
        n = len(offs) - 1
        out = np.zeros((n, self.cfg.max_candidates), dtype=cand_dtype)
        L = lib()
        rc = L.mc_partset_classify(self.h, seqs.ctypes.data, offs.ctypes.data, None, None, n, lowest, insert_max, out.ctypes.data)
        if rc != 0:
            raise McError(f"mc_partset_classify: {L.mc_partset_last_error(self.h).decode()} (rc {rc})")
        return out

    def close(self):
        if self.h:
            lib().mc_partset_close(self.h)
            self.h = C.c_void_p()


class KeySet:
    """ONE database key-sharded over the GPUs of the node (mc_keyset_*, Mode K from C++): every shard looks up its own features for all
    reads, the partial lists travel as 4-byte global window numbers to the shard that owns the read (RCCL between devices)."""

    def __init__(self, name: str, shards: int = 0, devices=None, **kw):
        L = lib()
        L.mc_keyset_open.argtypes = [C.c_char_p, C.POINTER(McConfig), C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)]
        L.mc_keyset_close.argtypes = [C.c_void_p]
        L.mc_keyset_info.argtypes = [C.c_void_p, C.c_void_p]
        L.mc_keyset_classify.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_uint64, C.c_void_p]
        L.mc_keyset_last_error.argtypes = [C.c_void_p]
        L.mc_keyset_last_error.restype = C.c_char_p
        self.cfg = default_config(**kw)
        self.h = C.c_void_p()
        dv = np.asarray(devices if devices is not None else [], dtype=np.int32)
        rc = L.mc_keyset_open(name.encode(), C.byref(self.cfg), shards, dv.ctypes.data if len(dv) else None, len(dv), C.byref(self.h))
        if rc != 0:
            raise McError(f"mc_keyset_open({name}): {L.mc_keyset_last_error(None).decode()} (rc {rc})")

    def info(self) -> dict:
        a = (C.c_uint64 * 8)()
        lib().mc_keyset_info(self.h, a)
        return dict(shards=int(a[0]), devices=int(a[1]), rccl=bool(a[2]), locations=int(a[3]), numbers_sent=int(a[4]), batches=int(a[5]),
                    reads_filtered=int(a[6]), locations_sorted=int(a[7]))

    def classify(self, reads, mates=None, lowest: int = 0, insert_max: int = 0) -> np.ndarray:
        """reads / mates: lists of bytes -> cand_dtype [n, max_candidates]"""
        def pack(rs):
            offs = np.zeros(len(rs) + 1, dtype=np.uint64)
            offs[1:] = np.cumsum([len(r) for r in rs])
            return np.frombuffer(b"".join(rs) + b"\0", dtype=np.uint8), offs
        n = len(reads)
        s1, o1 = pack(reads)
        s2, o2 = pack(mates) if mates is not None else (None, None)
        out = np.zeros((n, self.cfg.max_candidates), dtype=cand_dtype)
        L = lib()
        rc = L.mc_keyset_classify(self.h, s1.ctypes.data, o1.ctypes.data, s2.ctypes.data if s2 is not None else None,
                                  o2.ctypes.data if o2 is not None else None, n, lowest, insert_max, out.ctypes.data)
        if rc != 0:
            raise McError(f"mc_keyset_classify: {L.mc_keyset_last_error(self.h).decode()} (rc {rc})")
        return out

    def classify_packed(self, seqs: np.ndarray, offs: np.ndarray, lowest: int = 0, insert_max: int = 0) -> np.ndarray:
        """single-end reads as one byte array + n + 1 offsets (what mc_keyset_classify takes) -> cand_dtype [n, max_candidates]"""
        seqs = np.ascontiguousarray(seqs, dtype=np.uint8); offs = np.ascontiguousarray(offs, dtype=np.uint64)
        n = len(offs) - 1
        out = np.zeros((n, self.cfg.max_candidates), dtype=cand_dtype)
        L = lib()
        rc = L.mc_keyset_classify(self.h, seqs.ctypes.data, offs.ctypes.data, None, None, n, lowest, insert_max, out.ctypes.data)
        if rc != 0:
            raise McError(f"mc_keyset_classify: {L.mc_keyset_last_error(self.h).decode()} (rc {rc})")
        return out

    def close(self):
        if self.h:
            lib().mc_keyset_close(self.h)
            self.h = C.c_void_p()


class Builder:
    """Minimal database builder (mc_build_*): sketches targets on the GPU, writes reference-format files."""

    def __init__(self, **kw):
        self.cfg = default_config(**kw)
        self.h = C.c_void_p()
        rc = lib().mc_build_begin(C.byref(self.cfg), C.byref(self.h))
        if rc != MC_OK:
            raise McError(f"mc_build_begin -> {rc}: {lib().mc_last_error(None).decode()}")
        lib().mc_build_last_error.restype = C.c_char_p
        lib().mc_build_last_error.argtypes = [C.c_void_p]

    def _check(self, rc):
        if rc < 0:
            raise McError(f"builder error {rc}: {lib().mc_build_last_error(self.h).decode()}")

    def add_target(self, seq: np.ndarray | bytes, name: str, parent_taxid: int = 0, filename: str = ""):
        a = np.frombuffer(seq, dtype=np.uint8) if isinstance(seq, (bytes, bytearray)) else np.ascontiguousarray(seq, dtype=np.uint8)
        self._check(lib().mc_build_add_target(self.h, a.ctypes.data_as(C.c_void_p), a.size, name.encode(), parent_taxid, filename.encode()))

    def add_target_device(self, ptr: int, length: int, name: str, parent_taxid: int = 0, filename: str = "", file_index: int = 0):
        """a target whose characters are already in device memory (mc_build_add_target_device): valid until flush() / finish()"""
        L = lib()
        L.mc_build_add_target_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_char_p, C.c_int64, C.c_char_p, C.c_uint64]
        self._check(L.mc_build_add_target_device(self.h, ptr, length, name.encode(), parent_taxid, filename.encode(), file_index))

    def flush(self):
        lib().mc_build_flush.argtypes = [C.c_void_p]
        self._check(lib().mc_build_flush(self.h))

    def reserve(self, pairs: int):
        lib().mc_build_reserve.argtypes = [C.c_void_p, C.c_uint64]
        self._check(lib().mc_build_reserve(self.h, pairs))

    def table_begin(self, expect_keys: int = 0, expect_values: int = 0) -> "Database":
        """streaming table build (mc_build_table_begin): -> Database whose table takes finished builders through table_add()"""
        self._sync_cfg()
        out = C.c_void_p()
        lib().mc_build_table_begin.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_void_p)]
        self._check(lib().mc_build_table_begin(self.h, expect_keys, expect_values, C.byref(out)))
        cfg = McConfig.from_buffer_copy(self.cfg)
        cfg.key_shard_index, cfg.key_shard_count = 0, 1
        return Database.from_handle(out.value, cfg)

    def table_add(self, db: "Database"):
        lib().mc_build_table_add.argtypes = [C.c_void_p, C.c_void_p]
        self._check(lib().mc_build_table_add(db.h, self.h))

    @staticmethod
    def table_end(db: "Database"):
        lib().mc_build_table_end.argtypes = [C.c_void_p]
        db._check(lib().mc_build_table_end(db.h))
        db.refresh_info()

    def finish(self, load: bool = True, **query_kw) -> "Database | None":
        """Sort + bucketise.  load=True also returns a query Database holding the table."""
        if not load:
            self._check(lib().mc_build_finish(self.h, None))
            return None
        for k, v in query_kw.items():
            setattr(self.cfg, k, v)
        # the builder creates the query context from ITS config; update it first
        out = C.c_void_p()
        self._sync_cfg()
        self._check(lib().mc_build_finish(self.h, C.byref(out)))
        return Database.from_handle(out.value, self.cfg)

    @staticmethod
    def finish_shards(builders: "list[Builder]") -> "Database":
        """One query table from builders that were given the same targets and the key shards 0 .. n-1 of n (mc_build_finish_shards)."""
        for b in builders:
            b._sync_cfg()
        arr = (C.c_void_p * len(builders))(*[b.h for b in builders])
        out = C.c_void_p()
        lib().mc_build_finish_shards.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_void_p)]
        builders[0]._check(lib().mc_build_finish_shards(arr, len(builders), C.byref(out)))
        cfg = McConfig.from_buffer_copy(builders[0].cfg)
        cfg.key_shard_index, cfg.key_shard_count = 0, 1
        return Database.from_handle(out.value, cfg)

    def counts(self) -> tuple[int, int]:
        """(features, locations) held after finish (mc_build_counts)"""
        k, v = C.c_uint64(), C.c_uint64()
        lib().mc_build_counts.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        self._check(lib().mc_build_counts(self.h, C.byref(k), C.byref(v)))
        return k.value, v.value

    def remove_ambiguous(self, ancestor_of_target: np.ndarray, max_ambig: int = 1) -> int:
        """-remove-ambig-features after finish(load=False): ancestor_of_target[t] = id of target t's taxon on the chosen rank (0 = none).
        Returns the number of features dropped (mc_build_remove_ambiguous)."""
        a = np.ascontiguousarray(ancestor_of_target, dtype=np.uint32)
        rem = C.c_uint64()
        lib().mc_build_remove_ambiguous.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.POINTER(C.c_uint64)]
        self._check(lib().mc_build_remove_ambiguous(self.h, a.ctypes.data_as(C.c_void_p), a.size, max_ambig, C.byref(rem)))
        return rem.value

    def _sync_cfg(self):
        lib().mc_build_set_query_config.argtypes = [C.c_void_p, C.POINTER(McConfig)]
        lib().mc_build_set_query_config(self.h, C.byref(self.cfg))

    def write(self, name: str, taxa: list[tuple[int, int, int, str]]):
        """taxa: (id, parent, rank, name) of the non-target taxa"""
        class Rec(C.Structure):
            _fields_ = [("id", C.c_int64), ("parent", C.c_int64), ("rank", C.c_uint32), ("name", C.c_char_p)]
        arr = (Rec * max(len(taxa), 1))()
        keep = []
        for i, (tid, par, rk, nm) in enumerate(taxa):
            b = nm.encode(); keep.append(b)
            arr[i] = Rec(tid, par, rk, b)
        self._check(lib().mc_build_write(self.h, name.encode(), C.cast(arr, C.c_void_p), len(taxa)))

    @staticmethod
    def write_shards(builders: "list[Builder]", name: str, taxa: list[tuple[int, int, int, str]]):
        """<name>.meta + <name>.cache0 from the finished builders of one key-sharded set (mc_build_write_shards)"""
        class Rec(C.Structure):
            _fields_ = [("id", C.c_int64), ("parent", C.c_int64), ("rank", C.c_uint32), ("name", C.c_char_p)]
        arr = (Rec * max(len(taxa), 1))()
        keep = []
        for i, (tid, par, rk, nm) in enumerate(taxa):
            b = nm.encode(); keep.append(b)
            arr[i] = Rec(tid, par, rk, b)
        hs = (C.c_void_p * len(builders))(*[b.h for b in builders])
        lib().mc_build_write_shards.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.c_char_p, C.c_void_p, C.c_uint64]
        builders[0]._check(lib().mc_build_write_shards(hs, len(builders), name.encode(), C.cast(arr, C.c_void_p), len(taxa)))

    def write_begin(self, name: str, taxa: list[tuple[int, int, int, str]]):
        """streaming writer (mc_build_write_begin): -> handle for write_add / write_end"""
        class Rec(C.Structure):
            _fields_ = [("id", C.c_int64), ("parent", C.c_int64), ("rank", C.c_uint32), ("name", C.c_char_p)]
        arr = (Rec * max(len(taxa), 1))()
        keep = []
        for i, (tid, par, rk, nm) in enumerate(taxa):
            b = nm.encode(); keep.append(b)
            arr[i] = Rec(tid, par, rk, b)
        w = C.c_void_p()
        lib().mc_build_write_begin.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]
        self._check(lib().mc_build_write_begin(self.h, name.encode(), C.cast(arr, C.c_void_p), len(taxa), C.byref(w)))
        return w

    def write_add(self, writer):
        lib().mc_build_write_add.argtypes = [C.c_void_p, C.c_void_p]
        self._check(lib().mc_build_write_add(writer, self.h))

    @staticmethod
    def write_end(writer):
        lib().mc_build_write_end.argtypes = [C.c_void_p]
        rc = lib().mc_build_write_end(writer)
        if rc < 0:
            raise McError(f"mc_build_write_end -> {rc}")

    def free(self):
        if self.h:
            lib().mc_build_free(self.h)
            self.h = None
