"""CHECKER side of the RefSeq-scale workloads (tests and bench.py's parity / cpu_baseline legs only)."""
from __future__ import annotations

import numpy as np

import cpuref
from metacache_amd import synthdb


def oracle_database(spec: "synthdb.Phylogeny", wanted_features: np.ndarray | None, threads: int, with_lineages: bool = False, **sk):
    """The oracle's own restatement of the database build (oracle/mc_oracle.c: mco_db_build) over the same synthetic collection,
    restricted to the given features (None = all).  -> cpuref.CpuDb"""
    cs = synthdb.CpuSynth()
    T = np.ascontiguousarray(spec.targets)
    lin = None
    if with_lineages:
        n = len(T)
        lin = np.zeros((n, 21), dtype=np.int64)
        lin[:, 4] = 1000 + spec.species
        lin[:, 6] = 2_000_000 + spec.genus
    # a thread takes whole genera (consecutive targets with one ancestor): the generator then derives species and strains from cached codes
    g = spec.genus
    claim = int(np.bincount(g).max()) if len(g) else 1
    db = cpuref.oracle().build_db(T["length"], cs.target_callback(), T.ctypes.data, wanted=wanted_features, lineage=lin, threads=threads,
                                  claim=claim, **sk)
    db._keep = (cs, T)
    return db


def effective_cpus() -> int:
    """host threads worth using: the cgroup CPU quota where there is one (the GPU boxes show 256 CPUs and grant 16), else the CPU count"""
    import os
    n = os.cpu_count() or 1
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = max(1, min(n, int(round(int(q) / int(p)))))
    except (OSError, ValueError):
        pass
    return n


def sample_features(reads: list[bytes]) -> np.ndarray:
    """distinct features of a read sample (oracle sketcher, default parameters)"""
    orc = cpuref.oracle()
    feats = []
    for r in reads:
        f, c = orc.sketch(r)
        for w in range(len(c)):
            feats.append(f[w, :c[w]])
    return np.unique(np.concatenate(feats)) if feats else np.zeros(0, np.uint32)


# Every shipped variant of the lane path's kernels (mc_set_tuning switches; include/metacache_amd.h): each must give the results the
# default gives -- and those are compared with the oracle / the reference.  name -> switches set on top of the defaults.
VARIANTS = {
    "lane_fusion_quad": {"lane_fusion": 1, "quad_lookup": 1},       # sketch_probe_lane_kernel<true>
    "lane_fusion_private": {"lane_fusion": 1, "quad_lookup": 0},    # sketch_probe_lane_kernel<false>
    "apart": {"lane_fusion": 0, "quad_lookup": 0},                  # sketch_lane + probe_cands<false> + gw_filter_count
    "apart_quad_unfused_count": {"lane_fusion": 0, "quad_lookup": 1, "gw_fuse": 0},   # ... probe_cands<true>, gw_filter + gw_count apart
    "lane_fusion_five_waves": {"lane_fusion": 1, "gw_fuse": 5},      # gw_filter_count_kernel<.., 4>: 512 kept numbers in LDS, five waves per SIMD (rounds 4-5)
    "lane_fusion_six_waves": {"lane_fusion": 1, "gw_fuse": 6},       # gw_filter_count_kernel<.., 6>: 384 kept numbers, round table and distinct slots in LDS of their own (the default is <.., 7>)
    "direct_index_fused": {"direct_index": 1, "lane_fusion": 1},    # sketch_probe_lane_kernel<false, true>: lookups in the direct-address index (32 GiB beside the buckets)
    "direct_index_apart": {"direct_index": 1, "lane_fusion": 0},    # sketch_lane + probe_cands<false, true>
}
_DEFAULTS = {"lane_fusion": -1, "quad_lookup": -1, "gw_fuse": 1, "direct_index": -1}


def each_variant(db, names=None):
    """generator: sets one variant's switches on the live context, yields its name, restores the defaults afterwards"""
    for name, sw in VARIANTS.items():
        if names and name not in names:
            continue
        for k, v in {**_DEFAULTS, **sw}.items():
            db.set_tuning(k, v)
        try:
            yield name
        finally:
            for k, v in _DEFAULTS.items():
                db.set_tuning(k, v)
