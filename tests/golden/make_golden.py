#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ FROM THE REFERENCE ITSELF.

Runs only in the build container (needs oracle/_ref = the reference compiled from /root/reference by
`make -C oracle ref`).  What it writes is data only:

  toy32.meta/.cache0        database built by the reference CLI (`metacache build`, uint32 target ids)
  toy16.meta/.cache0        same genomes, reference built with MC_TARGET_ID_TYPE=uint16_t (6-byte locations)
  toy32p2.meta/.cache{0,1}  same genomes, `-parts 2`
  toy_reads.npz             the query reads (single, paired, long, edge cases)
  toy32_expected.npz ...    per read: the reference's allhits + top candidates for several rule sets,
                            obtained in-process from database::query_host through oracle/ref_shim.cpp
  sketch_vectors.npz        strings -> window sketches from the reference's sketcher

Usage:  python tests/golden/make_golden.py
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
import tarfile
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from metacache_amd import synth  # noqa: E402
import cpuref  # noqa: E402

REF_TAX = "/root/reference/test/taxonomy.tar.gz"

# (name, max_cand, lowest_rank, insert_size_max) -- rule sets recorded for single-end reads
SINGLE_RULES = [("c2_seq", 2, 0, 0), ("c3_species", 3, 4, 0), ("call_seq", 0, 0, 0), ("call_genus", 0, 6, 0),
                ("c1_seq", 1, 0, 0)]
PAIR_RULES = [("c2_seq", 2, 0, 0), ("c2_seq_ins700", 2, 0, 700), ("c4_species", 4, 4, 0)]


def make_genomes(rng):
    """8 species x 3 strains (3 % divergence) of 16 kbp + a 448-bp repeat copied to window-aligned
    positions so that some buckets exceed the 254-location cap."""
    species = synth.TOY_SPECIES[:8]
    repeat = synth.random_genome(rng, 448)
    genomes, headers = [], []
    acc = 1
    for sp in species:
        base = synth.random_genome(rng, 16000)
        for strain in range(3):
            g = synth.mutate(rng, base, 0.03)
            # 15 aligned repeat copies per genome -> 24*15 = 360 > 254 copies of each repeat window
            for slot in rng.choice(np.arange(1, 138), size=15, replace=False):
                p = int(slot) * 112
                g[p:p + 448] = repeat[: min(448, g.size - p)]
            genomes.append(g)
            headers.append(f"NC_{acc:06d}.1 kraken:taxid|{sp} synthetic species {sp} strain {strain}")
            acc += 1
    return genomes, headers, repeat


def make_reads(rng, genomes, repeat):
    single = []
    r, _, _ = synth.sample_reads(rng, genomes, 1500, 150, 0.01, 0.002)
    single += [bytes(x) for x in r]
    single += [bytes(synth.random_genome(rng, 150)) for _ in range(100)]
    g0 = genomes[0]
    for L in (0, 1, 15, 16, 17, 20, 31, 100, 126, 127, 128, 129, 142, 143, 150, 238, 239, 240, 254, 255, 351, 352):
        st = int(rng.integers(0, g0.size - max(L, 1)))
        single.append(bytes(g0[st:st + L]))
    single.append(bytes(g0[500:650]).lower())
    single.append(bytes(g0[500:650]).replace(b"T", b"U"))
    single.append(bytes(g0[700:850]).lower().replace(b"t", b"u"))
    single.append(b"N" * 150)
    x = bytearray(g0[900:1050]); x[15::16] = b"N" * len(x[15::16]); single.append(bytes(x))
    x = bytearray(g0[900:1050]); x[60] = ord("R"); x[61] = ord("-"); x[100] = ord("n"); single.append(bytes(x))
    single.append(bytes(repeat[:150]))
    single.append(bytes(repeat[100:400]))
    single.append(bytes(synth.revcomp(repeat)[20:170]))
    single.append(b"A" * 150)
    single.append(b"ACGT" * 40)
    # long reads 200..4000 bp with 5 % substitutions / 0.5 % N
    for _ in range(120):
        L = int(min(4000, max(200, rng.lognormal(np.log(480), 0.8))))
        g = genomes[int(rng.integers(0, len(genomes)))]
        st = int(rng.integers(0, g.size - L))
        s = g[st:st + L]
        if rng.random() < 0.5:
            s = synth.revcomp(s)
        single.append(bytes(synth.mutate(rng, s, 0.05, 0.005)))
    # pairs: fragment 300..500, mate 2 reverse-complemented
    p1, p2 = [], []
    for _ in range(400):
        g = genomes[int(rng.integers(0, len(genomes)))]
        F = int(rng.integers(300, 501))
        st = int(rng.integers(0, g.size - F))
        frag = g[st:st + F]
        if rng.random() < 0.5:
            frag = synth.revcomp(frag)
        p1.append(bytes(synth.mutate(rng, frag[:150], 0.01, 0.002)))
        p2.append(bytes(synth.mutate(rng, synth.revcomp(frag)[:150], 0.01, 0.002)))
    p1 += [b"", bytes(g0[100:250]), b"N" * 150]
    p2 += [bytes(g0[100:250]), b"", bytes(g0[4000:4150])]
    return single, p1, p2


def run(cmd):
    print("+", " ".join(cmd))
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def expected_for(db, single, p1, p2):
    out = {}
    all_hits, all_off = [], [0]
    for s in single:
        h, _ = db.query(s, b"", 2, 0, 0)
        all_hits.append(h); all_off.append(all_off[-1] + len(h))
    out["single_allhits"] = np.concatenate(all_hits) if all_hits else np.zeros(0, cpuref.hit_dtype)
    out["single_allhits_off"] = np.array(all_off, dtype=np.uint64)
    for name, mc, low, ins in SINGLE_RULES:
        c_list, c_off = [], [0]
        for s in single:
            _, c = db.query(s, b"", mc, low, ins)
            c_list.append(c); c_off.append(c_off[-1] + len(c))
        out[f"single_{name}"] = np.concatenate(c_list)
        out[f"single_{name}_off"] = np.array(c_off, dtype=np.uint64)
    all_hits, all_off = [], [0]
    for a, b in zip(p1, p2):
        h, _ = db.query(a, b, 2, 0, 0)
        all_hits.append(h); all_off.append(all_off[-1] + len(h))
    out["pair_allhits"] = np.concatenate(all_hits)
    out["pair_allhits_off"] = np.array(all_off, dtype=np.uint64)
    for name, mc, low, ins in PAIR_RULES:
        c_list, c_off = [], [0]
        for a, b in zip(p1, p2):
            _, c = db.query(a, b, mc, low, ins)
            c_list.append(c); c_off.append(c_off[-1] + len(c))
        out[f"pair_{name}"] = np.concatenate(c_list)
        out[f"pair_{name}_off"] = np.array(c_off, dtype=np.uint64)
    out["lineages"] = db.lineages()
    out["target_names"] = np.array([db.target_name(t) for t in range(db.n_targets)])
    out["info"] = np.array(db.info(), dtype=np.uint64)
    return out


def main():
    if not (cpuref.have_reference(4) and cpuref.have_reference(2)):
        sys.exit("oracle/_ref missing: run `make -C oracle ref` first")
    rng = np.random.default_rng(20240928)
    genomes, headers, repeat = make_genomes(rng)
    single, p1, p2 = make_reads(rng, genomes, repeat)

    tmp = tempfile.mkdtemp(prefix="mcgold")
    try:
        with tarfile.open(REF_TAX) as tf:
            tf.extractall(tmp)
        fdir = os.path.join(tmp, "fa")
        os.makedirs(fdir)
        for i, (h, g) in enumerate(zip(headers, genomes)):
            synth.write_fasta(os.path.join(fdir, f"g{i:02d}.fa"), [(h, g)])
        tax = os.path.join(tmp, "taxonomy")
        u32 = os.path.join(ROOT, "oracle", "_ref", "metacache_u32")
        u16 = os.path.join(ROOT, "oracle", "_ref", "metacache_u16")
        run([u32, "build", os.path.join(tmp, "toy32"), fdir, "-taxonomy", tax])
        run([u16, "build", os.path.join(tmp, "toy16"), fdir, "-taxonomy", tax])
        run([u32, "build", os.path.join(tmp, "toy32p2"), fdir, "-taxonomy", tax, "-parts", "2"])
        for f in os.listdir(tmp):
            if f.endswith(".meta") or ".cache" in f:
                shutil.copy(os.path.join(tmp, f), os.path.join(HERE, f))
    finally:
        shutil.rmtree(tmp)

    sb, so = synth.pack_reads(single)
    b1, o1 = synth.pack_reads(p1)
    b2, o2 = synth.pack_reads(p2)
    np.savez_compressed(os.path.join(HERE, "toy_reads.npz"), single=sb, single_off=so, p1=b1, p1_off=o1, p2=b2, p2_off=o2)

    for name, tb in (("toy32", 4), ("toy16", 2)):
        db = cpuref.reference(tb).open(os.path.join(HERE, name))
        out = expected_for(db, single, p1, p2)
        # query-time table modifiers on a fresh handle
        db.set_max_locations_per_feature(2)
        sub = single[:200] + single[1600:1640]
        c_list, c_off, h_list, h_off = [], [0], [], [0]
        for s in sub:
            h, c = db.query(s, b"", 2, 0, 0)
            h_list.append(h); h_off.append(h_off[-1] + len(h)); c_list.append(c); c_off.append(c_off[-1] + len(c))
        out["maxloc2_idx"] = np.array(list(range(200)) + list(range(1600, 1640)))
        out["maxloc2_allhits"] = np.concatenate(h_list); out["maxloc2_allhits_off"] = np.array(h_off, dtype=np.uint64)
        out["maxloc2_c2_seq"] = np.concatenate(c_list); out["maxloc2_c2_seq_off"] = np.array(c_off, dtype=np.uint64)
        db.close()
        db = cpuref.reference(tb).open(os.path.join(HERE, name))
        out["rmover_removed"] = np.array([db.remove_features_with_more_locations_than(3)])
        c_list, c_off, h_list, h_off = [], [0], [], [0]
        for s in sub:
            h, c = db.query(s, b"", 2, 0, 0)
            h_list.append(h); h_off.append(h_off[-1] + len(h)); c_list.append(c); c_off.append(c_off[-1] + len(c))
        out["rmover_allhits"] = np.concatenate(h_list); out["rmover_allhits_off"] = np.array(h_off, dtype=np.uint64)
        out["rmover_c2_seq"] = np.concatenate(c_list); out["rmover_c2_seq_off"] = np.array(c_off, dtype=np.uint64)
        db.close()
        np.savez_compressed(os.path.join(HERE, f"{name}_expected.npz"), **out)

    # multi-part: the in-process reference result is history dependent, so the
    # reads are run strictly in order on ONE handler and recorded in that order.
    db = cpuref.reference(4).open(os.path.join(HERE, "toy32p2"))
    out = {}
    h_list, h_off, c_list, c_off = [], [0], [], [0]
    for s in single:
        h, c = db.query(s, b"", 2, 0, 0)
        h_list.append(h); h_off.append(h_off[-1] + len(h)); c_list.append(c); c_off.append(c_off[-1] + len(c))
    out["single_allhits"] = np.concatenate(h_list); out["single_allhits_off"] = np.array(h_off, dtype=np.uint64)
    out["single_c2_seq"] = np.concatenate(c_list); out["single_c2_seq_off"] = np.array(c_off, dtype=np.uint64)
    out["lineages"] = db.lineages()
    out["info"] = np.array(db.info(), dtype=np.uint64)
    db.close()
    np.savez_compressed(os.path.join(HERE, "toy32p2_expected.npz"), **out)

    # sketch vectors
    ref = cpuref.reference(4)
    strings = single[1600:1640] + [bytes(genomes[1][:1000]), bytes(genomes[2][3000:3400]).lower()]
    params = [(16, 16, 127, 112), (16, 8, 127, 112), (12, 16, 64, 53), (16, 16, 127, 50), (16, 4, 40, 60), (7, 32, 100, 94),
              (16, 32, 255, 240)]
    sk = {}
    sb2, so2 = synth.pack_reads(strings)
    sk["strings"] = sb2; sk["strings_off"] = so2; sk["params"] = np.array(params, dtype=np.uint32)
    for pi, (k, s, w, st) in enumerate(params):
        feats, counts, nw = [], [], []
        for x in strings:
            f, c = ref.sketch(x, k, s, w, st)
            feats.append(f.reshape(-1)); counts.append(c); nw.append(len(c))
        sk[f"p{pi}_feats"] = np.concatenate(feats); sk[f"p{pi}_counts"] = np.concatenate(counts)
        sk[f"p{pi}_nwin"] = np.array(nw, dtype=np.uint32)
    np.savez_compressed(os.path.join(HERE, "sketch_vectors.npz"), **sk)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
