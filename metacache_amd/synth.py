"""Seeded synthetic genomes and reads (no network, so every DB / read set is generated).

Pure numpy; used by the golden-fixture generator, the tests and bench.py.  Nothing here is on the
measured path.
"""
from __future__ import annotations

import numpy as np

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.zeros(256, dtype=np.uint8)
_COMP[:] = ord("N")
for a, b in zip(b"ACGTUacgtuN", b"TGCAAtgcaaN"):
    _COMP[a] = b

# species-level taxids present in the reference's toy taxonomy (test/taxonomy.tar.gz)
TOY_SPECIES = (562, 813, 29459, 37372, 2151, 74109, 56, 9, 336810, 1048758, 91844, 1160784)


def random_genome(rng: np.random.Generator, length: int) -> np.ndarray:
    """i.i.d. uniform ACGT, as uint8 ASCII."""
    return ACGT[rng.integers(0, 4, size=length, dtype=np.uint8)]


def mutate(rng: np.random.Generator, seq: np.ndarray, sub_rate: float, n_rate: float = 0.0) -> np.ndarray:
    """Substitutions (to a *different* base) at sub_rate, then 'N' at n_rate."""
    out = seq.copy()
    if sub_rate > 0:
        m = rng.random(out.size) < sub_rate
        idx = np.nonzero(m)[0]
        if idx.size:
            # shift base by 1..3 inside ACGT; non-ACGT characters become a random base
            code = np.searchsorted(ACGT, out[idx]) % 4
            out[idx] = ACGT[(code + rng.integers(1, 4, size=idx.size)) % 4]
    if n_rate > 0:
        out[rng.random(out.size) < n_rate] = ord("N")
    return out


def revcomp(seq: np.ndarray) -> np.ndarray:
    return _COMP[seq[::-1]]


def sample_reads(rng: np.random.Generator, genomes: list[np.ndarray], n: int, length: int = 150,
                 sub_rate: float = 0.01, n_rate: float = 0.001, both_strands: bool = True):
    """n fixed-length reads: uniform genome, uniform start, random strand, substitutions, N's.

    Returns (reads[n, length] uint8, genome_index[n], start[n]).
    """
    g = rng.integers(0, len(genomes), size=n)
    reads = np.empty((n, length), dtype=np.uint8)
    starts = np.empty(n, dtype=np.int64)
    for gi in range(len(genomes)):
        sel = np.nonzero(g == gi)[0]
        if sel.size == 0:
            continue
        G = genomes[gi]
        st = rng.integers(0, G.size - length + 1, size=sel.size)
        starts[sel] = st
        reads[sel] = G[st[:, None] + np.arange(length)[None, :]]
    if both_strands:
        flip = rng.random(n) < 0.5
        reads[flip] = _COMP[reads[flip][:, ::-1]]
    flat = mutate(rng, reads.reshape(-1), sub_rate, n_rate)
    return flat.reshape(n, length), g, starts


def pack_reads(reads) -> tuple[np.ndarray, np.ndarray]:
    """list of byte-like reads (or a 2-D uint8 array) -> (concatenated uint8, offsets uint64[n+1])."""
    if isinstance(reads, np.ndarray) and reads.ndim == 2:
        n, L = reads.shape
        return np.ascontiguousarray(reads).reshape(-1), (np.arange(n + 1, dtype=np.uint64) * np.uint64(L))
    lens = np.array([len(r) for r in reads], dtype=np.uint64)
    offs = np.zeros(len(reads) + 1, dtype=np.uint64)
    np.cumsum(lens, out=offs[1:])
    buf = np.empty(int(offs[-1]), dtype=np.uint8)
    for r, o in zip(reads, offs[:-1]):
        buf[int(o):int(o) + len(r)] = np.frombuffer(bytes(r), dtype=np.uint8) if not isinstance(r, np.ndarray) else r
    return buf, offs


def write_fasta(path: str, records: list[tuple[str, np.ndarray]], width: int = 80) -> None:
    with open(path, "wb") as f:
        for header, seq in records:
            f.write(b">" + header.encode() + b"\n")
            b = seq.tobytes()
            for i in range(0, len(b), width):
                f.write(b[i:i + width] + b"\n")
