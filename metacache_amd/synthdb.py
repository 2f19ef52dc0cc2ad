"""Synthetic RefSeq-shaped workloads (BASELINE.json configs[2..4]) -- WORKLOAD GENERATION ONLY.

A collection is a table of `syn_target` records (metacache_amd/synth/synth_spec.h): every base of every target is a pure function
of (target record, position), so 150 Gbp of targets never have to exist anywhere as a whole: the GPU generator writes the targets a
builder pass needs into HBM, the CPU generator hands the same characters to the oracle, and reads are drawn by evaluating the
function -- no genome is resident while the table is queried.

    spec = phylogeny(genera=2000, species_per_genus=4, strains_per_species=5, len_min=2_500_000, len_max=5_000_000, seed=3100)
    gen = GpuSynth(device); gen.targets(spec, first, count, dst_tensor)        # ASCII into HBM
    gen.reads(spec, params, first_read, n, dst_tensor)                         # [n, row_bytes] uint8
    CpuSynth().target(spec, t) / .reads(...)                                   # the same bytes on the host
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import build as _build

target_dtype = np.dtype([("genus_seed", "<u8"), ("species_seed", "<u8"), ("strain_seed", "<u8"), ("thr_species", "<u4"),
                         ("thr_strain", "<u4"), ("length", "<u4"), ("pad", "<u4")])
assert target_dtype.itemsize == 40


class ReadParams(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("read_len", C.c_uint32), ("row_bytes", C.c_uint32), ("thr_sub", C.c_uint32), ("thr_n", C.c_uint32),
                ("paired", C.c_uint32), ("frag_min", C.c_uint32), ("frag_max", C.c_uint32), ("num_targets", C.c_uint32)]


def _thr(p: float) -> int:
    return min(int(p * 4294967296.0), 0xFFFFFFFF)


def read_params(spec: "Phylogeny", seed: int, read_len: int = 150, sub_rate: float = 0.01, n_rate: float = 0.001, paired: bool = False,
                frag_min: int = 300, frag_max: int = 500) -> ReadParams:
    row = (read_len + 3) // 4 * 4
    if row == read_len:
        row += 4                                   # every row ends in zero bytes (and starts 4-byte aligned)
    return ReadParams(seed, read_len, row, _thr(sub_rate), _thr(n_rate), int(paired), frag_min, frag_max, len(spec.targets))


@dataclass
class Phylogeny:
    targets: np.ndarray          # [n] target_dtype
    species: np.ndarray          # [n] species index of each target
    genus: np.ndarray            # [n] genus index of each target
    n_species: int
    n_genera: int

    @property
    def total_bases(self) -> int:
        return int(self.targets["length"].astype(np.int64).sum())

    def offsets(self, first: int, count: int) -> np.ndarray:
        """4-byte aligned start of each of the targets [first, first + count) in one buffer, plus the total size (count + 1 entries)"""
        ln = (self.targets["length"][first:first + count].astype(np.int64) + 3) // 4 * 4
        off = np.zeros(count + 1, dtype=np.uint64)
        off[1:] = np.cumsum(ln)
        return off

    def subset(self, sel: np.ndarray) -> "Phylogeny":
        """the targets sel (ascending) as a collection of their own -- one PART of a partitioned database; taxa keep their numbers"""
        sel = np.asarray(sel, dtype=np.int64)
        return Phylogeny(np.ascontiguousarray(self.targets[sel]), self.species[sel].copy(), self.genus[sel].copy(), self.n_species, self.n_genera)

    def lineages(self) -> np.ndarray:
        """[targets, 21] taxon index + 1 per rank for mc_set_lineages: sequence level = the target itself, rank 4 = species,
        rank 6 = genus (taxonomy.hpp:68-91); taxon indices: targets first, then species, then genera"""
        n = len(self.targets)
        lin = np.zeros((n, 21), dtype=np.uint32)
        lin[:, 0] = np.arange(n) + 1
        lin[:, 4] = n + 1 + self.species
        lin[:, 6] = n + self.n_species + 1 + self.genus
        return lin

    def taxa(self):
        """(id, parent, rank, name) of the non-target taxa for Builder.write: root, genera, species"""
        out = [(1, 1, 20, "root")]
        out += [(2_000_000 + g, 1, 6, f"synthetic genus {g}") for g in range(self.n_genera)]
        sp_genus = np.zeros(self.n_species, dtype=np.int64)
        sp_genus[self.species] = self.genus
        out += [(1000 + s, 2_000_000 + int(sp_genus[s]), 4, f"synthetic species {s}") for s in range(self.n_species)]
        return out

    def parent_taxid(self, t: int) -> int:
        return 1000 + int(self.species[t])


def phylogeny(genera: int, species_per_genus: int, strains_per_species: int, len_min: int, len_max: int, seed: int,
              div_species=(0.04, 0.10), div_strain=(0.005, 0.02), big_fraction: float = 0.0, big_len=(0, 0)) -> Phylogeny:
    """genus -> species -> strain.  All members of a genus have the genus' length (substitutions only); a species differs from the
    genus ancestor in div_species of its bases, strain 0 of a species IS the species sequence, the others differ from it in
    div_strain of their bases (uniform in the given ranges, per species / per strain)."""
    rng = np.random.default_rng(seed)
    n = genera * species_per_genus * strains_per_species
    T = np.zeros(n, dtype=target_dtype)
    sp = np.zeros(n, dtype=np.int64)
    ge = np.zeros(n, dtype=np.int64)
    gseed = rng.integers(1, 1 << 63, size=genera, dtype=np.uint64)
    glen = rng.integers(len_min, len_max + 1, size=genera, dtype=np.int64)
    sseed = rng.integers(1, 1 << 63, size=genera * species_per_genus, dtype=np.uint64)
    sdiv = rng.uniform(div_species[0], div_species[1], size=genera * species_per_genus)
    tseed = rng.integers(1, 1 << 63, size=n, dtype=np.uint64)
    tdiv = rng.uniform(div_strain[0], div_strain[1], size=n)
    if big_fraction > 0:                                       # a few genera of LARGE genomes (drawn from a stream of their own: the defaults stay as they were)
        r2 = np.random.default_rng(seed + 0x5EED)
        big = r2.random(genera) < big_fraction
        glen = np.where(big, r2.integers(big_len[0], big_len[1] + 1, size=genera, dtype=np.int64), glen)
    i = 0
    for g in range(genera):
        for s in range(species_per_genus):
            si = g * species_per_genus + s
            for k in range(strains_per_species):
                T[i] = (gseed[g], sseed[si], tseed[i], _thr(sdiv[si]), 0 if k == 0 else _thr(tdiv[i]), glen[g], 0)
                sp[i] = si; ge[i] = g
                i += 1
    return Phylogeny(T, sp, ge, genera * species_per_genus, genera)


class CpuSynth:
    def __init__(self):
        _, path = _build.build_synth()
        self.lib = C.CDLL(path)
        self.lib.mcs_cpu_target.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        self.lib.mcs_cpu_reads.argtypes = [C.POINTER(ReadParams), C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]
        self.lib.mcs_cpu_origins.argtypes = [C.POINTER(ReadParams), C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]

    def target(self, spec: Phylogeny, t: int, first: int = 0, n: int | None = None) -> np.ndarray:
        rec = np.ascontiguousarray(spec.targets[t:t + 1])
        n = int(rec["length"][0]) - first if n is None else n
        out = np.empty(n, dtype=np.uint8)
        self.lib.mcs_cpu_target(rec.ctypes.data, first, n, out.ctypes.data)
        return out

    def target_callback(self):
        """(function address, keep-alive) for the oracle's restricted database build: void cb(void* targets, u32 t, char* dst)"""
        return C.cast(self.lib.mcs_cpu_target_cb, C.c_void_p).value

    def reads(self, spec: Phylogeny, P: ReadParams, first: int, n: int):
        """-> rows[n, row_bytes] (and mate-2 rows when paired)"""
        T = np.ascontiguousarray(spec.targets)
        a = np.empty((n, P.row_bytes), dtype=np.uint8)
        b = np.empty((n, P.row_bytes), dtype=np.uint8) if P.paired else None
        self.lib.mcs_cpu_reads(C.byref(P), T.ctypes.data, first, n, a.ctypes.data, None if b is None else b.ctypes.data)
        return (a, b) if P.paired else a

    def origins(self, spec: Phylogeny, P: ReadParams, first: int, n: int) -> np.ndarray:
        """[n, 4] = target, start, reverse strand, fragment length: the ground truth of the reads"""
        T = np.ascontiguousarray(spec.targets)
        out = np.empty((n, 4), dtype=np.uint32)
        self.lib.mcs_cpu_origins(C.byref(P), T.ctypes.data, first, n, out.ctypes.data)
        return out


class GpuSynth:
    """Writes targets / reads into torch CUDA tensors (torch is plumbing: device memory)."""

    def __init__(self, device=0):
        import torch
        self.torch = torch
        self.dev = torch.device("cuda", device) if isinstance(device, int) else device
        path, _ = _build.build_synth()
        self.lib = C.CDLL(path)
        self.lib.mcs_targets.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        self.lib.mcs_reads.argtypes = [C.POINTER(ReadParams), C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
        self._spec_id = None
        self._dT = None

    def _targets_dev(self, spec: Phylogeny):
        if self._spec_id != id(spec):
            self._dT = self.torch.from_numpy(np.ascontiguousarray(spec.targets).view(np.uint8).copy()).to(self.dev)
            self._spec_id = id(spec)
        return self._dT

    def targets(self, spec: Phylogeny, first: int, count: int, dst) -> np.ndarray:
        """targets [first, first + count) as ASCII into the uint8 tensor dst (4-byte aligned starts); returns their offsets (count + 1)"""
        torch = self.torch
        off = spec.offsets(first, count)
        assert dst.dtype == torch.uint8 and dst.numel() >= int(off[-1]) + 16
        dT = self._targets_dev(spec)
        dOff = torch.from_numpy(off[:-1].astype(np.int64)).to(self.dev)
        stream = torch.cuda.current_stream(self.dev).cuda_stream
        rc = self.lib.mcs_targets(dT.data_ptr() + first * 40, dOff.data_ptr(), count, dst.data_ptr(), stream)
        if rc:
            raise RuntimeError(f"mcs_targets -> {rc}")
        torch.cuda.current_stream(self.dev).synchronize()       # dOff may go away
        return off

    def reads(self, spec: Phylogeny, P: ReadParams, first: int, n: int, dst, dst2=None):
        torch = self.torch
        assert dst.dtype == torch.uint8 and dst.numel() >= n * P.row_bytes
        dT = self._targets_dev(spec)
        stream = torch.cuda.current_stream(self.dev).cuda_stream
        rc = self.lib.mcs_reads(C.byref(P), dT.data_ptr(), first, n, dst.data_ptr(), dst2.data_ptr() if dst2 is not None else None, stream)
        if rc:
            raise RuntimeError(f"mcs_reads -> {rc}")


def build_database(spec: Phylogeny, device: int = 0, shards: int = 1, chunk_bytes: int = 3 << 30, report=None, key_shard=(0, 1), write_to: str | None = None,
                   only_targets: np.ndarray | None = None, **cfg):
    """The collection as a query table in HBM, built by the product's builder (mc_build_*) from targets that are generated on the
    device group by group: `shards` key-shard passes (every pass sketches all targets and keeps 1/shards of the features, sorts them
    and inserts them into the table: mc_build_table_*), so that neither the targets (150 Gbp) nor all (feature, location) pairs
    (2 x 10^10) ever exist at once.  key_shard = (i, n): only the features of key shard i of n (mc_key_owner) -- one rank's table in
    Mode K; its `shards` build passes are the sub-shards i * shards .. i * shards + shards - 1 of n * shards.
    write_to: also write the database as files <write_to>.meta / .cache0 in the reference's format, shard by shard (mc_build_write_*).
    only_targets: ONE PART of a partitioned database -- all targets keep their numbers and stand in the metadata, but only these
    (ascending) are sketched; the others go in as names and window counts (mc_build_add_existing_target).  tools/partgroup_bench.py.
    cfg: Builder / mc_config fields (max_candidates, max_load_factor, target_id_bytes, ...).
    -> (api.Database, info dict with seconds per phase)"""
    import time
    import torch
    from . import api
    dev = torch.device("cuda", device)
    gen = GpuSynth(device)
    n = len(spec.targets)
    lens = spec.targets["length"].astype(np.int64)
    # groups of consecutive targets that fit the generation buffer
    groups, first, acc = [], 0, 0
    for t in range(n):
        ln = (int(lens[t]) + 3) // 4 * 4
        if acc + ln > chunk_bytes and t > first:
            groups.append((first, t - first)); first, acc = t, 0
        acc += ln
    groups.append((first, n - first))
    buf_bytes = max(int(spec.offsets(f, c)[-1]) for f, c in groups) + 64
    buf = torch.zeros(buf_bytes, dtype=torch.uint8, device=dev)
    cfg.setdefault("target_id_bytes", 4)
    stride = cfg.get("winstride", 112) or 112
    sk = cfg.get("sketchlen", 16) or 16
    est_pairs = int((lens // stride + 2).sum()) * sk
    info = {"targets": n, "bases": int(lens.sum()), "shards": shards, "seconds": {"generate": 0.0, "sketch": 0.0, "sort": 0.0, "insert": 0.0}}
    names = [f"SYN_{t:06d}.1".encode() for t in range(n)]
    db = None
    writer = None
    L = api.lib()
    mine = None
    if only_targets is not None:
        mine = np.zeros(n, dtype=bool); mine[np.asarray(only_targets, dtype=np.int64)] = True
        est_pairs = int((lens[mine] // stride + 2).sum()) * sk
        L.mc_build_add_existing_target.argtypes = [C.c_void_p, C.c_char_p, C.c_int64, C.c_char_p, C.c_uint64, C.c_uint64]
        kk = cfg.get("kmerlen", 16) or 16
        wl = cfg.get("winlen", 127) or 127
    L.mc_build_add_target_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_char_p, C.c_int64, C.c_char_p, C.c_uint64]
    t_all = time.time()
    for sh in range(shards):
        kw = dict(cfg)
        if shards * key_shard[1] > 1:
            kw.update(key_shard_index=key_shard[0] * shards + sh, key_shard_count=key_shard[1] * shards)
        b = api.Builder(device=device, **kw)
        b.reserve(int(est_pairs / (shards * key_shard[1]) * 1.02) + (1 << 20))
        for f, c in groups:
            t0 = time.time()
            off = gen.targets(spec, f, c, buf)
            torch.cuda.synchronize(dev)
            t1 = time.time()
            base = buf.data_ptr()
            for i in range(c):
                t = f + i
                if mine is not None and not mine[t]:
                    # another part's target: its place in the numbering and its window count (hash_dna.hpp:54-75), no sequence
                    ln = int(lens[t])
                    if ln <= wl:
                        nw = 1 if ln >= kk else 0
                    else:
                        nw = (ln - wl) // stride + 1
                        if nw * stride < ln and ln - nw * stride >= kk:
                            nw += 1
                    rc = L.mc_build_add_existing_target(b.h, names[t], 1000 + int(spec.species[t]), b"", 0, nw)
                else:
                    rc = L.mc_build_add_target_device(b.h, base + int(off[i]), int(lens[t]), names[t], 1000 + int(spec.species[t]), b"", 0)
                if rc < 0:
                    b._check(rc)
            b.flush()
            t2 = time.time()
            info["seconds"]["generate"] += t1 - t0; info["seconds"]["sketch"] += t2 - t1
        t0 = time.time()
        b.finish(load=False)
        t1 = time.time()
        if db is None:
            k0, v0 = b.counts()                                   # the passes of one table are alike: keys are dealt out by a hash
            db = b.table_begin(int(k0 * shards * 1.02) + (1 << 16), int(v0 * shards * 1.04) + (1 << 20)) if shards > 1 else b.table_begin(max(k0, 1), max(v0, 1))
        b.table_add(db)
        t2 = time.time()
        if write_to:
            if writer is None:
                writer = b.write_begin(write_to, spec.taxa())
            b.write_add(writer)
            info["seconds"]["write_files"] = info["seconds"].get("write_files", 0.0) + time.time() - t2
        info["seconds"]["sort"] += t1 - t0; info["seconds"]["insert"] += t2 - t1
        if report:
            report(f"shard {sh + 1}/{shards}: {b.counts()} (features, locations), {time.time() - t_all:.1f} s")
        b.free()
    api.Builder.table_end(db)
    if writer is not None:
        api.Builder.write_end(writer)
    del buf
    torch.cuda.empty_cache()
    info["seconds"] = {k: round(v, 2) for k, v in info["seconds"].items()}
    info["seconds"]["total"] = round(time.time() - t_all, 2)
    return db, info
