#!/usr/bin/env python3
"""Golden vectors for the command line: read files + the reference CLI's own output for them.

Runs only in the build container (needs oracle/_ref/metacache_u32 = the reference compiled from
/root/reference by `make -C oracle ref`, and the toy32 database written by make_golden.py).  Writes data only:

  cli_reads.fa                  400 single-end reads (multi-line FASTA, headers with descriptions, edge cases)
  cli_pairs.fq                  120 pairs, interleaved FASTQ (for -pairseq)
  cli_p1.fa / cli_p2.fa         the same pairs as two FASTA files (for -pairfiles)
  cli_expected.json.gz           {case: {"args": [...], "files": [...], "lines": [...]}}: the complete output file the
                                reference's `metacache query toy32 <files> <args> -threads 1` wrote (-out)

Usage:  python tests/golden/make_golden_cli.py
"""
from __future__ import annotations

import gzip
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref", "metacache_u32")

# name -> (input files, options)
CASES = {
    "default": (["cli_reads.fa"], []),
    "everything_species": (["cli_reads.fa"], ["-queryids", "-tophits", "-allhits", "-lineage", "-taxids", "-lowest", "species"]),
    "allhits_sequence": (["cli_reads.fa"], ["-allhits", "-tophits", "-maxcand", "3"]),
    "genus_family_idsonly": (["cli_reads.fa"], ["-lowest", "genus", "-highest", "family", "-taxids-only", "-omit-ranks"]),
    "separate_cols": (["cli_reads.fa"], ["-separate-cols", "-lineage", "-lowest", "species", "-highest", "order", "-taxids"]),
    "mapped_only_vote": (["cli_reads.fa"], ["-mapped-only", "-hitmin", "5", "-hitdiff", "0.5", "-maxcand", "4", "-tophits"]),
    "hitdiff_percent": (["cli_reads.fa"], ["-hitdiff", "80", "-maxcand", "3", "-lowest", "species", "-tophits", "-queryids"]),
    "maxcand_unlimited": (["cli_reads.fa"], ["-maxcand", "0", "-tophits", "-lowest", "subspecies"]),
    "maxcand_unlimited_seq": (["cli_reads.fa", "cli_pairs.fq"], ["-maxcand", "0", "-tophits", "-queryids"]),
    "separator": (["cli_reads.fa"], ["-separator", ";", "-taxids", "-lineage", "-highest", "genus"]),
    "pairseq": (["cli_pairs.fq"], ["-pairseq", "-tophits", "-queryids"]),
    "pairseq_insert": (["cli_pairs.fq"], ["-pairseq", "-insertsize", "700", "-tophits", "-lowest", "species"]),
    "pairfiles": (["cli_p2.fa", "cli_p1.fa"], ["-pairfiles", "-tophits", "-allhits", "-queryids"]),
    "two_files": (["cli_reads.fa", "cli_pairs.fq"], ["-queryids", "-taxids"]),
    "sketch_params": (["cli_reads.fa"], ["-sketchlen", "12", "-winlen", "100", "-winstride", "80", "-tophits"]),
    "max_locations": (["cli_reads.fa"], ["-max-locations-per-feature", "4", "-tophits"]),
    "remove_overpopulated": (["cli_reads.fa"], ["-remove-overpopulated-features", "-max-locations-per-feature", "20", "-tophits"]),
    "gzip_input": (["cli_reads.fa.gz"], ["-tophits", "-queryids"]),
    "directory_input": (["cli_dir"], ["-queryids"]),
    "readlen_filter": (["cli_reads.fa"], ["-min-readlen", "100", "-max-readlen", "300", "-queryids"]),
    "readlen_filter_limit": (["cli_reads.fa", "cli_pairs.fq"], ["-min-readlen", "140", "-query-limit", "60", "-queryids"]),
    "query_limit": (["cli_reads.fa", "cli_pairs.fq"], ["-query-limit", "50", "-queryids", "-pairseq"]),
    "comment_token": (["cli_reads.fa"], ["-comment", "%%", "-mapped-only"]),
    "locations": (["cli_reads.fa"], ["-locations", "-maxcand", "3"]),
    "locations_pairs": (["cli_pairs.fq"], ["-pairseq", "-locations", "-insertsize", "600"]),
    "abundances": (["cli_reads.fa"], ["-abundances"]),
    "abundances_species": (["cli_reads.fa", "cli_pairs.fq"], ["-abundances", "-abundance-per", "species", "-lowest", "subspecies"]),
    "abundance_per_genus_nomap": (["cli_reads.fa"], ["-no-map", "-abundance-per", "genus"]),
    "abundance_per_sequence": (["cli_pairs.fq"], ["-pairseq", "-no-map", "-abundance-per", "sequence", "-hitdiff", "50", "-maxcand", "4"]),
    "hits_per_ref": (["cli_reads.fa"], ["-no-map", "-hits-per-ref"]),
    "hits_per_ref_lineage": (["cli_pairs.fq"], ["-pairseq", "-hits-per-ref", "-lineage", "-highest", "genus", "-taxids", "-maxcand", "3"]),
    "ground_truth": (["cli_truth.fa"], ["-ground-truth", "-taxids"]),
    "precision": (["cli_truth.fa"], ["-precision", "-mapped-only", "-lowest", "species"]),
    "precision_truth_lineage": (["cli_truth.fa"], ["-precision", "-ground-truth", "-lineage", "-separate-cols", "-tophits"]),
    # input the strict 4-line / no-'+' fast scans cannot take: the reader's sequential state machine decides (sequence_io.cpp:160-236)
    "fastq_irregular": (["cli_irregular.fq"], ["-queryids", "-tophits"]),
    "fastq_irregular_pairseq": (["cli_irregular.fq"], ["-pairseq", "-queryids"]),
    "fasta_plus_lines": (["cli_irregular.fa"], ["-queryids", "-tophits"]),
    "irregular_with_regular": (["cli_irregular.fa", "cli_pairs.fq", "cli_irregular.fq"], ["-queryids"]),
    # -cov-percentile: targets with the lowest coverage are dropped, the reads re-classified (classification.cpp:591-634, :747-838)
    "cov_percentile": (["cli_reads.fa"], ["-cov-percentile", "0.3", "-tophits", "-queryids"]),
    "cov_percentile_pct_hits_per_ref": (["cli_reads.fa", "cli_pairs.fq"], ["-cov-percentile", "55", "-hits-per-ref", "-maxcand", "3", "-batch-size", "100"]),
    "cov_percentile_species": (["cli_pairs.fq"], ["-pairseq", "-cov-percentile", "0.5", "-lowest", "species", "-abundances", "-tophits"]),
    "reference_test_matrix": (["cli_truth.fa"], ["-mapped-only", "-precision", "-ground-truth", "-tophits", "-allhits", "-abundances", "-abundance-per", "species"]),
}

# name -> (input files, options with {targets} / {abund} standing for extra output files)
EXTRA_FILE_CASES = {
    "analysis_files": (["cli_reads.fa"], ["-hits-per-ref", "{targets}", "-abundances", "{abund}", "-abundance-per", "family", "-mapped-only"]),
}

# name -> (input files, options incl. '-split-out'): one output file per input (pair)
SPLIT_CASES = {
    "split_out": (["cli_reads.fa", "cli_pairs.fq"], ["-queryids"]),
    "split_out_pairfiles": (["cli_p1.fa", "cli_p2.fa"], ["-pairfiles", "-tophits"]),
}

# interactive mode: lines typed at the prompt (each gets its own -out file); defaults from the initial options
INTERACTIVE = {
    "initial": ["-tophits"],
    "lines": [["cli_reads.fa", "-queryids"], ["cli_pairs.fq", "-pairseq", "-lowest", "species"], ["cli_p1.fa", "cli_p2.fa", "-pairfiles", "-maxcand", "3"]],
}


def format_matrix():
    """the option matrix of the reference's own formatting test (test/run_tests:86-117): 3 x 4 x 12 = 144 combinations"""
    lines = []
    for outer in ("", "-mapped-only", "-separator /%/"):
        for mid in ("", "-omit-ranks", "-queryids", "-queryids -omit-ranks"):
            for tax in ("", "-taxids", "-taxids-only"):
                for lay in ("", "-lineage", "-separate-cols", "-lineage -separate-cols"):
                    lines.append(("-no-summary -no-query-params " + " ".join(x for x in (outer, mid, lay, tax) if x)).split())
    return lines


def wrap(seq: bytes, width: int) -> str:
    s = seq.decode()
    return "\n".join(s[i:i + width] for i in range(0, len(s), width)) if s else ""


def main():
    if not os.path.exists(REF):
        sys.exit("oracle/_ref missing: run `make -C oracle ref` first")
    z = np.load(os.path.join(HERE, "toy_reads.npz"))

    def unpack(b, o):
        return [bytes(b[int(o[i]):int(o[i + 1])]) for i in range(len(o) - 1)]
    single, p1, p2 = unpack(z["single"], z["single_off"]), unpack(z["p1"], z["p1_off"]), unpack(z["p2"], z["p2_off"])
    pick = list(range(0, 240)) + list(range(1500, 1640)) + list(range(1640, 1660))     # sampled, random, edge cases, long
    with open(os.path.join(HERE, "cli_reads.fa"), "w") as f:
        for n, i in enumerate(pick):
            hdr = f"read{n:04d}" if n % 3 else f"read{n:04d} source=toy idx={i}"
            f.write(f">{hdr}\n")
            body = wrap(single[i], 60 if n % 2 else 100000)
            if body:
                f.write(body + "\n")
            if n % 50 == 7:
                f.write("\n")                                                           # blank lines are skipped
    # headers that carry a ground truth in the forms ground_truth() understands (classification.cpp:104-137): accession.version of
    # a target, accession without version, taxid|<id>, a target name as leading word, nothing usable
    taxids = [562, 813, 2151, 9, 56, 29459, 37372, 74109, 99999]
    with open(os.path.join(HERE, "cli_truth.fa"), "w") as f:
        for n in range(300):
            kind = n % 6
            acc = f"NC_{(n * 7) % 24 + 1:06d}"
            if kind == 0: hdr = f"t{n:04d} {acc}.1 sampled"
            elif kind == 1: hdr = f"t{n:04d}|{acc}|frag"
            elif kind == 2: hdr = f"t{n:04d} taxid|{taxids[n % len(taxids)]}|x"
            elif kind == 3: hdr = f"{acc}.1"
            elif kind == 4: hdr = f"t{n:04d} nothing here"
            else: hdr = f"gi|12345|ref|{acc}.2| taxid {taxids[(n // 6) % len(taxids)]}"
            f.write(f">{hdr}\n{single[n].decode()}\n")
    with open(os.path.join(HERE, "cli_pairs.fq"), "w") as f:
        for n in range(120):
            for m, s in ((1, p1[n]), (2, p2[n])):
                f.write(f"@pair{n:03d}/{m} len={len(s)}\n{s.decode()}\n+\n{'I' * len(s)}\n")
    with open(os.path.join(HERE, "cli_irregular.fq"), "w") as f:
        for n in range(60):
            sq = single[n + 700].decode()
            kind = n % 10
            if kind == 0:                                                           # sequence over three lines
                f.write(f"@irr{n:02d} multi\n{sq[:50]}\n{sq[50:100]}\n{sq[100:]}\n+\n{'I' * len(sq)}\n")
            elif kind == 1:                                                         # blank line inside the sequence
                f.write(f"@irr{n:02d} blank\n{sq[:70]}\n\n{sq[70:]}\n+irr{n:02d}\n{'F' * len(sq)}\n")
            elif kind == 2:                                                         # qualities over two lines: the second one is a stray line
                f.write(f"@irr{n:02d} q2\n{sq}\n+\n{'I' * 75}\n{'I' * 75}\n")
            elif kind == 3:                                                         # stray lines between records
                f.write(f"stray line\n\n@irr{n:02d} stray\n{sq}\n+\n{'I' * len(sq)}\n# another one\n")
            elif kind == 4:                                                         # a FASTA record among FASTQ records, two lines
                f.write(f">irr{n:02d} fasta\n{sq[:80]}\n{sq[80:]}\n")
            elif kind == 5:                                                         # no sequence at all
                f.write(f"@irr{n:02d} empty\n+\n\n")
            elif kind == 6:                                                         # quality line that begins with '@' followed by a strict record
                f.write(f"@irr{n:02d} atq\n{sq}\n+\n@{'I' * (len(sq) - 1)}\n")
            elif kind == 7:                                                         # Windows line ends
                f.write(f"@irr{n:02d} crlf\r\n{sq}\r\n+\r\n{'I' * len(sq)}\r\n")
            else:
                f.write(f"@irr{n:02d} plain\n{sq}\n+\n{'I' * len(sq)}\n")
        f.write(f"@irr_last no quality, no newline\n{single[777].decode()}")
    with open(os.path.join(HERE, "cli_irregular.fa"), "w") as f:
        for n in range(40):
            sq = single[n + 800].decode()
            if n % 8 == 3:                                                          # a '+' line ends the record, the line after it is dropped
                f.write(f">pl{n:02d} plus\n{sq[:90]}\n+\n{sq[90:]}\n")
            elif n % 8 == 5:                                                        # FASTQ record inside a FASTA file
                f.write(f"@pl{n:02d} fastq\n{sq}\n+\n{'I' * len(sq)}\n")
            elif n % 8 == 6:                                                        # sequence line that begins with '@' is data
                f.write(f">pl{n:02d} at\n{sq[:60]}\n@{sq[60:]}\n")
            else:
                f.write(f">pl{n:02d}\n{wrap(single[n + 800], 70)}\n")
    for name, mates in (("cli_p1.fa", p1), ("cli_p2.fa", p2)):
        with open(os.path.join(HERE, name), "w") as f:
            for n in range(120):
                f.write(f">pair{n:03d}\n{mates[n].decode()}\n")

    with open(os.path.join(HERE, "cli_reads.fa"), "rb") as f, open(os.path.join(HERE, "cli_reads.fa.gz"), "wb") as raw:
        with gzip.GzipFile(fileobj=raw, mode="wb", mtime=0) as g:
            g.write(f.read())
    os.makedirs(os.path.join(HERE, "cli_dir", "sub"), exist_ok=True)
    with open(os.path.join(HERE, "cli_dir", "sub", "nested_reads.fa"), "w") as f:      # one file, two levels down
        for n in range(40):
            f.write(f">nested{n:03d}\n{single[n + 300].decode()}\n")

    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for name, (files, args) in CASES.items():
            res = os.path.join(tmp, name + ".txt")
            cmd = [REF, "query", "toy32"] + files + args + ["-threads", "1", "-out", res]
            print("+", " ".join(cmd))
            subprocess.check_call(cmd, cwd=HERE, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            with open(res) as f:
                out[name] = {"files": files, "args": args, "lines": f.read().split("\n")}
        for name, (files, args) in EXTRA_FILE_CASES.items():
            res = os.path.join(tmp, name + ".txt")
            extra = {"targets": os.path.join(tmp, name + ".targets"), "abund": os.path.join(tmp, name + ".abund")}
            cmd = [REF, "query", "toy32"] + files + [a.format(**extra) for a in args] + ["-threads", "1", "-out", res]
            print("+", " ".join(cmd))
            subprocess.check_call(cmd, cwd=HERE, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            rec = {"files": files, "args": args, "main": open(res).read().split("\n"), "extra": {}}
            for k, fn in extra.items():
                rec["extra"][k] = open(fn).read().split("\n")
            out[name] = rec
        for name, (files, args) in SPLIT_CASES.items():
            prefix = os.path.join(tmp, name)
            cmd = [REF, "query", "toy32"] + files + args + ["-threads", "1", "-split-out", prefix]
            print("+", " ".join(cmd))
            subprocess.check_call(cmd, cwd=HERE, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            outs = {}
            for fn in sorted(os.listdir(tmp)):
                if fn.startswith(name + "_"):
                    with open(os.path.join(tmp, fn)) as f:
                        outs[fn[len(name):]] = f.read().split("\n")
            out[name] = {"files": files, "args": args, "split": outs}
        stdin = ""
        for i, line in enumerate(INTERACTIVE["lines"]):
            stdin += " ".join(line + ["-out", os.path.join(tmp, f"inter{i}.txt")]) + "\n"
        cmd = [REF, "query", "toy32"] + INTERACTIVE["initial"] + ["-threads", "1"]
        print("+", " ".join(cmd), "<<", repr(stdin))
        subprocess.run(cmd, cwd=HERE, input=stdin + "\n", text=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
        outs = []
        for i in range(len(INTERACTIVE["lines"])):
            with open(os.path.join(tmp, f"inter{i}.txt")) as f:
                outs.append(f.read().split("\n"))
        out["interactive"] = {"initial": INTERACTIVE["initial"], "lines": INTERACTIVE["lines"], "outputs": outs}
        # the reference's formatting matrix, one interactive session, 30 reads
        with open(os.path.join(HERE, "cli_fmt.fa"), "w") as f:
            for n in range(30):
                f.write(f">fmt{n:02d} NC_{n % 24 + 1:06d}.1\n{single[n * 11].decode()}\n")
        matrix = format_matrix()
        stdin = ""
        for i, line in enumerate(matrix):
            stdin += " ".join(["cli_fmt.fa"] + line + ["-out", os.path.join(tmp, f"fmt{i}.txt")]) + "\n"
        subprocess.run([REF, "query", "toy32", "-threads", "1"], cwd=HERE, input=stdin + "\n", text=True, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL, check=True)
        outs = []
        for i in range(len(matrix)):
            with open(os.path.join(tmp, f"fmt{i}.txt")) as f:
                outs.append(f.read().split("\n"))
        out["format_matrix"] = {"matrix": matrix, "outputs": outs}
    with gzip.open(os.path.join(HERE, "cli_expected.json.gz"), "wt") as f:
        json.dump(out, f)
    print("wrote", len(out), "cases")


if __name__ == "__main__":
    main()
