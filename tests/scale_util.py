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
    db = cpuref.oracle().build_db(T["length"], cs.target_callback(), T.ctypes.data, wanted=wanted_features, lineage=lin, threads=threads, **sk)
    db._keep = (cs, T)
    return db


def sample_features(reads: list[bytes]) -> np.ndarray:
    """distinct features of a read sample (oracle sketcher, default parameters)"""
    orc = cpuref.oracle()
    feats = []
    for r in reads:
        f, c = orc.sketch(r)
        for w in range(len(c)):
            feats.append(f[w, :c[w]])
    return np.unique(np.concatenate(feats)) if feats else np.zeros(0, np.uint32)
