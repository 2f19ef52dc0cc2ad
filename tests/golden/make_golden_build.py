#!/usr/bin/env python3
"""Golden vectors for `build` / `build+query`: reference sequence files, taxonomy dumps and id tables, and what the REFERENCE
made of them.

Runs only in the build container (needs oracle/_ref/metacache_u32 and metacache_u16 = the reference compiled from
/root/reference by `make -C oracle ref`).  Writes data only, under tests/golden/:

  build_in/genomes/...            reference sequences: multi-line FASTA, NCBI / gi / plain headers, taxid in header, duplicate id,
                                  empty record, a sequence shorter than k, lower case + N, an assembly_summary.txt beside them
  build_in/taxonomy/              nodes.dmp / names.dmp / merged.dmp of the reference's own test (test/taxonomy.tar.gz) + an
                                  .accession2taxid table for the ranking pass after the build
  build_reads.fa                  reads drawn from those sequences
  build_expected.json.gz          per case: the taxa and the feature -> locations map of the database files the reference's
                                  `metacache build` wrote (parsed, order-free), and the output files of `metacache query` on it
                                  and of `metacache build+query`

Usage:  python tests/golden/make_golden_build.py
"""
from __future__ import annotations

import gzip
import json
import os
import shutil
import struct
import subprocess
import sys
import tarfile
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF32 = os.path.join(ROOT, "oracle", "_ref", "metacache_u32")
REF16 = os.path.join(ROOT, "oracle", "_ref", "metacache_u16")
REF_TAX = "/root/reference/test/taxonomy.tar.gz"
IN = os.path.join(HERE, "build_in")

# files are named explicitly and in this order (a directory would be walked in readdir order, which differs between machines);
# the last argument is a directory with ONE file
FILES = ["build_in/genomes/GCF_000001111.1_ASM111v1_genomic.fna", "build_in/genomes/mixed.fa", "build_in/genomes/assembly_summary.txt",
         "build_in/genomes/more.fa.gz", "build_in/genomes/sub"]
TAX = ["-taxonomy", "build_in/taxonomy"]

# name -> (reference binary, extra build options, mcq's extra build options)
BUILD_CASES = {
    "default": (REF32, [], []),
    "u16": (REF16, [], ["-target-id-type", "uint16_t"]),
    "overpopulated": (REF32, ["-max-locations-per-feature", "6", "-remove-overpopulated-features"], None),
    "maxlocs": (REF32, ["-max-locations-per-feature", "3"], None),
    "leadingword": (REF32, ["-sequence-id-format", "leadingword"], None),
    "sketching": (REF32, ["-kmerlen", "12", "-sketchlen", "8", "-winlen", "64", "-winstride", "40"], None),
    "no_taxonomy": (REF32, None, None),                      # built without -taxonomy
    "reset_taxa": (REF32, ["-reset-taxa"], None),
    "ambig_species": (REF32, ["-remove-ambig-features", "species"], None),
    "ambig_genus_2": (REF32, ["-remove-ambig-features", "genus", "-max-ambig-per-feature", "2"], None),
    "ambig_sequence_3": (REF32, ["-remove-ambig-features", "sequence", "-max-ambig-per-feature", "3", "-max-locations-per-feature", "20"], None),
}
# modify: name -> (binary, files + options of the first build, files + options of `modify`, mcq's extra options for both)
MODIFY_CASES = {
    # modify takes no -threads: the generator pins the reference to one CPU so that target ids follow the file order.  The reference
    # parses the command line twice (options.cpp:772,786) and so adds every new file TWICE, the second time under '<id>!1' names.
    "modify_default": (REF32, FILES[:4] + TAX, FILES[4:], []),
    "modify_adds_taxonomy": (REF32, FILES[:4], FILES[4:] + TAX, []),
    # (no case with a -max-locations-per-feature below 254 in the first build: the reference's modify then lets the buckets it read from
    #  the file grow past that limit -- 6 + 6 locations under a header that says 6 -- which mcq does not imitate: it keeps the first n)
    "modify_u16": (REF16, FILES[:3] + FILES[4:] + TAX, FILES[3:4], ["-target-id-type", "uint16_t"]),
    "modify_ambig": (REF32, FILES[:3] + FILES[4:] + TAX, FILES[3:4] + TAX + ["-remove-ambig-features", "genus", "-reset-taxa"], []),
}
QUERY_ARGS = ["-tophits", "-allhits", "-queryids", "-lowest", "species", "-taxids", "-lineage"]
# build+query has no thread option for its build half: with several input files the reference builds in several parts whose
# consumer threads take target ids in schedule order.  ONE input file => one part => reproducible ids.
BQ_FILES = ["build_in/genomes/mixed.fa"]
BQ_CASES = {
    "bq_default": ([], ["-tophits", "-queryids", "-taxids"]),
    "bq_species": (["-max-locations-per-feature", "6", "-remove-overpopulated-features"], ["-lowest", "species", "-tophits", "-allhits", "-maxcand", "3"]),
    "bq_save": (["-save-db", "{savedb}"], ["-tophits", "-lowest", "genus"]),
    "bq_ambig": (["-remove-ambig-features", "species"], ["-tophits", "-allhits", "-taxids"]),
}


def rc(a):
    comp = np.zeros(256, dtype=np.uint8)
    for x, y in zip(b"ACGTN", b"TGCAN"):
        comp[x] = y
    return comp[a[::-1]]


def mutate(rng, g, rate):
    g = g.copy()
    pos = np.nonzero(rng.random(g.size) < rate)[0]
    g[pos] = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=pos.size)
    return g


def fasta(records, width=70):
    out = []
    for h, s in records:
        out.append(">" + h)
        s = bytes(s).decode()
        if not s:
            continue
        for i in range(0, len(s), width):
            out.append(s[i:i + width])
    return "\n".join(out) + "\n"


def make_inputs():
    rng = np.random.default_rng(20260928)
    shutil.rmtree(IN, ignore_errors=True)
    os.makedirs(os.path.join(IN, "genomes", "sub"))
    os.makedirs(os.path.join(IN, "taxonomy"))
    with tarfile.open(REF_TAX) as t:
        for m in t.getmembers():
            if m.isfile():
                with open(os.path.join(IN, "taxonomy", os.path.basename(m.name)), "wb") as f:
                    f.write(t.extractfile(m).read())
    base = [rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=int(n)) for n in (24000, 18000, 30000, 15000)]
    repeat = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=336)
    for g in base[:3]:                                   # a repeat shared by several sequences: buckets beyond the small limits
        for p in (1120, 5600, 11200):
            g[p:p + 336] = repeat
    strain = lambda i, r: mutate(rng, base[i], r)
    lower = strain(1, 0.02).copy()
    lower[2000:2600] = np.frombuffer(bytes(lower[2000:2600]).lower(), dtype=np.uint8)
    lower[4000:4040] = ord("N")
    genomes = {}
    recs1 = [("NC_000111.1 Escherichia coli strain one chromosome, complete genome", base[0]),
             ("NC_000112.1 Escherichia coli strain one plasmid pX", base[3][:6000])]
    open(os.path.join(IN, "genomes", "GCF_000001111.1_ASM111v1_genomic.fna"), "w").write(fasta(recs1, 80))
    recs2 = [("NZ_ABCD01000001.1 Chlamydia trachomatis contig 1, whole genome shotgun sequence", strain(0, 0.03)),
             ("gi|55555|gb|whatever Some organism with a genbank id", base[1]),
             ("plain_name_without_accession kraken:taxid|29459|", lower),
             ("NC_000111.1 a second sequence with an id that is taken", strain(0, 0.05)[:9000]),
             ("NC_000113.1 record without sequence", b""),
             ("NC_000114.1 shorter than k", base[2][:10]),
             ("NC_000115.1 one window exactly", base[2][100:227]),
             ("NC_000111.1 third use of the id", strain(2, 0.01)[:12000])]
    open(os.path.join(IN, "genomes", "mixed.fa"), "w").write(fasta(recs2, 60))
    recs3 = [("NC_000116.2 Brucella melitensis chromosome I taxid=29459 ", base[2]),
             ("AB123456.1 unranked sequence, no taxon anywhere", strain(3, 0.02))]
    with gzip.open(os.path.join(IN, "genomes", "more.fa.gz"), "wt") as f:
        f.write(fasta(recs3, 70))
    recs4 = [("NC_000117.1 Buchnera aphidicola", strain(3, 0.04)), ("NC_000118.1 Buchnera again", strain(1, 0.06))]
    open(os.path.join(IN, "genomes", "sub", "GCF_000002222.2_other.fa"), "w").write(fasta(recs4, 70))
    # NCBI-style table: the file GCF_000001111.1* -> 562; GCF_000002222.2 is found in the global table of the taxonomy directory
    open(os.path.join(IN, "genomes", "assembly_summary.txt"), "w").write(
        "#   See ftp://ftp.ncbi.nlm.nih.gov/genomes/README_assembly_summary.txt for a description of the columns in this file.\n"
        "# assembly_accession\tbioproject\tbiosample\twgs_master\trefseq_category\ttaxid\tspecies_taxid\torganism_name\n"
        "GCF_000001111.1\tPRJNA1\tSAMN1\t\treference genome\t562\t562\tEscherichia coli\n"
        "GCF_000009999.1\tPRJNA2\tSAMN2\t\tna\t813\t813\tChlamydia trachomatis\n")
    open(os.path.join(IN, "taxonomy", "assembly_summary_refseq.txt"), "w").write(
        "# assembly_accession\tbioproject\tbiosample\twgs_master\trefseq_category\ttaxid\tspecies_taxid\torganism_name\n"
        "GCF_000002222.2\tPRJNA3\tSAMN3\t\tna\t9\t9\tBuchnera aphidicola\n")
    open(os.path.join(IN, "taxonomy", "toy.accession2taxid"), "w").write(
        "accession\taccession.version\ttaxid\tgi\n"
        "NZ_ABCD01000001\tNZ_ABCD01000001.1\t813\t111\n"
        "XX_000000\tXX_000000.1\t2151\t55555\n"
        "NC_000114\tNC_000114.7\t74109\t333\n"
        "NC_000111\tNC_000111.1\t37372\t444\n")
    for recs in (recs1, recs2, recs3, recs4):
        for h, s in recs:
            genomes[h] = np.frombuffer(bytes(s), dtype=np.uint8)
    # reads
    pool = [g for g in genomes.values() if g.size > 2000]
    lines = []
    for i in range(300):
        g = pool[int(rng.integers(len(pool)))]
        L = int(rng.integers(60, 260))
        p = int(rng.integers(0, g.size - L))
        r = mutate(rng, np.frombuffer(bytes(g[p:p + L]).upper(), dtype=np.uint8), 0.02)
        if rng.random() < 0.5:
            r = rc(r)
        lines.append(f">read{i} len={L}\n{bytes(r).decode()}")
    open(os.path.join(HERE, "build_reads.fa"), "w").write("\n".join(lines) + "\n")


def parse_db(name):
    """order-free content of <name>.meta / .cache0 (database.cpp:247-290, hash_multimap.hpp:1037-1082)"""
    b = open(name + ".meta", "rb").read()
    p = 0

    def rd(fmt):
        nonlocal p
        v = struct.unpack_from("<" + fmt, b, p)
        p += struct.calcsize("<" + fmt)
        return v if len(v) > 1 else v[0]

    def rstr():
        nonlocal p
        n = rd("Q")
        s = b[p:p + n].decode()
        p += n
        return s
    ver = rd("Q")
    widths = rd("7B")
    sk1 = rd("4Q"); sk2 = rd("4Q")
    maxlocs = rd("Q")
    ntgt = rd("H") if widths[1] == 2 else rd("I")
    nparts = rd("I")
    ntaxa = rd("Q")
    taxa = {}
    for _ in range(ntaxa):
        tid, parent, rank = rd("q"), rd("q"), rd("B")
        nm, fn = rstr(), rstr()
        idx, win = rd("Q"), rd("Q")
        taxa[str(tid)] = [parent, rank, nm, fn, idx, win]
    assert p == len(b)
    c = open(name + ".cache0", "rb").read()
    nkeys, nvals, batch = struct.unpack_from("<3Q", c, 0)
    p = 24
    tb = widths[1]
    feats = {}
    done = 0
    while done < nkeys:
        nb = min(batch, nkeys - done)
        keys = np.frombuffer(c, dtype="<u4", count=nb, offset=p); p += 4 * nb
        sizes = np.frombuffer(c, dtype=np.uint8, count=nb, offset=p); p += nb
        for k, s in zip(keys.tolist(), sizes.tolist()):
            vals = []
            for _ in range(s):
                win = struct.unpack_from("<I", c, p)[0]
                tgt = struct.unpack_from("<H" if tb == 2 else "<I", c, p + 4)[0]
                p += 4 + tb
                vals.append([tgt, win])
            feats[str(k)] = vals
        done += nb
    assert p == len(c) and sum(len(v) for v in feats.values()) == nvals
    return {"version": ver, "widths": list(widths), "sketching": list(sk1), "sketching2": list(sk2), "maxlocs": maxlocs, "targets": ntgt,
            "parts": nparts, "taxa": taxa, "features": feats}


def run(cmd, **kw):
    r = subprocess.run(cmd, cwd=HERE, capture_output=True, text=True, timeout=600, **kw)
    if r.returncode != 0:
        sys.exit(f"FAILED: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
    return r


def main():
    for b in (REF32, REF16):
        if not os.path.exists(b):
            sys.exit("oracle/_ref is missing: run `make -C oracle ref` first")
    make_inputs()
    out = {"files": FILES, "tax": TAX, "query_args": QUERY_ARGS, "build": {}, "bq": {}}
    with tempfile.TemporaryDirectory() as tmp:
        for name, (ref, extra, mcq_extra) in BUILD_CASES.items():
            db = os.path.join(tmp, name)
            args = FILES + (TAX if extra is not None else []) + (extra or [])
            run([ref, "build", db] + args + ["-threads", "1"])
            res = os.path.join(tmp, name + ".txt")
            run([ref, "query", db, "build_reads.fa"] + QUERY_ARGS + ["-threads", "1", "-out", res])
            out["build"][name] = {"args": args, "mcq_extra": mcq_extra or [], "db": parse_db(db), "query": open(res).read().split("\n")}
        out["modify"] = {}
        for name, (ref, first, second, mcq_extra) in MODIFY_CASES.items():
            db = os.path.join(tmp, name)
            run([ref, "build", db] + first + ["-threads", "1"])
            run(["taskset", "-c", "0", ref, "modify", db] + second)      # one hardware thread = one build part: files in the given order
            res = os.path.join(tmp, name + ".txt")
            run([ref, "query", db, "build_reads.fa"] + QUERY_ARGS + ["-threads", "1", "-out", res])
            out["modify"][name] = {"first": first, "second": second, "mcq_extra": mcq_extra, "db": parse_db(db), "query": open(res).read().split("\n")}
        for name, (bargs, qargs) in BQ_CASES.items():
            res = os.path.join(tmp, name + ".txt")
            args = ["-targets"] + BQ_FILES + TAX + bargs + ["-query", "build_reads.fa"] + qargs
            saved = os.path.join(tmp, name + "_saved")
            run([REF32, "build+query"] + [a.format(savedb=saved) for a in args] + ["-threads", "1", "-out", res])
            out["bq"][name] = {"args": args, "lines": open(res).read().split("\n")}
            if os.path.exists(saved + ".meta"):
                out["bq"][name]["saved"] = parse_db(saved)
    with gzip.open(os.path.join(HERE, "build_expected.json.gz"), "wt") as f:
        json.dump(out, f)
    print("wrote build_expected.json.gz:", {k: (len(v["db"]["taxa"]), len(v["db"]["features"])) for k, v in out["build"].items()})


if __name__ == "__main__":
    main()
