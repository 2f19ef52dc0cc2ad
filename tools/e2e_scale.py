"""End-to-end `query` at RefSeq-like scale (kernel-only AND end-to-end): a cut of the bench collection (--scale 0.2: 8 000
targets, 30 Gbp, 4.3 x 10^9 locations) is built on the GPU and written as database files to /dev/shm by the streaming writer; 10^7
synthetic 150 bp reads go to a FASTA file; then `mcq query` (this repository, MI355X) and the reference's own command line
(oracle/_ref/metacache_u32, all granted host threads) run on the SAME files: database load time, query time, Mreads/min, and the
`-tophits -queryids` mapping lines of a sample diffed.  Never bench.py's `value`: parsing, PCIe, classification and printing included.

    python tools/e2e_scale.py --scale 0.2 --reads 10000000 --out gpurun_out/e2e_scale02.json"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import bench  # noqa: E402
import e2e_bench  # noqa: E402
from metacache_amd import build, synthdb  # noqa: E402


def cpu_stat():
    out = {}
    for fn in ("cpu.stat", "memory.stat", "memory.events"):
        try:
            out.update({k: int(v) for k, v in (l.split() for l in open("/sys/fs/cgroup/" + fn))})
        except Exception:
            pass
    return out


def timed(cmd, env=None):
    t0 = time.perf_counter()
    c0 = cpu_stat()
    r = subprocess.run(cmd, capture_output=True, text=True, env=env)
    c1 = cpu_stat()
    if r.returncode != 0:
        raise RuntimeError(" ".join(cmd) + "\n" + r.stderr[-2000:])
    prof = [l for l in r.stderr.splitlines() if l.startswith("mcq profile") or l.startswith("mc submit trace") or l.startswith("mc slot warm-up")]
    ens = [l for l in r.stderr.splitlines() if l.startswith("mc ensure")]        # (MC_ALLOC_TRACE=1: allocations of 20 ms and more)
    if ens:
        import re
        ms = [sum(float(x) for x in re.findall(r"([0-9.]+) ms", l)) for l in ens]
        prof.append(f"{len(ens)} workspace allocations traced, {sum(ms):.0f} ms in all; the largest: " + "; ".join(sorted(ens, key=lambda l: -sum(float(x) for x in re.findall(r"([0-9.]+) ms", l)))[:6]))
    if c0 and c1:
        prof.append("cgroup: " + ", ".join(f"{k} +{c1[k] - c0[k]}" for k in ("usage_usec", "nr_throttled", "pgfault", "pgmajfault", "pgscan", "pgsteal", "high", "max", "oom") if k in c0 and k in c1))
    return time.perf_counter() - t0, prof


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=0.2)
    ap.add_argument("--reads", type=int, default=10_000_000)
    ap.add_argument("--cpu-reads", type=int, default=400_000)
    ap.add_argument("--no-ref", action="store_true")
    ap.add_argument("--key-shards", type=int, default=0, help="also run `mcq query -shard keys -key-shards n` (the database as n key shards on this GPU, mc_keyset_*)")
    ap.add_argument("--batch-sizes", default="", help="comma list: the three mcq runs once more per -batch-size value (default run: mcq's own 65 536)")
    ap.add_argument("--pipes", default="", help="comma list: the -no-map and the -tophits run once more per MC_PIPES value (batches in flight on the device; default 8)")
    ap.add_argument("--repeat-nomap", type=int, default=0, help="the -no-map run this many times back to back, then as often with --sleep seconds before each (is the query phase's rate a property of the process' memory?)")
    ap.add_argument("--sleep", type=float, default=20.0)
    ap.add_argument("--profile-nomap", type=int, default=0, help="the -no-map run this many times under rocprofv3 --kernel-trace --stats: the kernels' total time beside the query phase's")
    ap.add_argument("--repeat-prefix", default="", help="';'-separated command prefixes (e.g. 'taskset -c 0-31;taskset -c 64-95'): the repeated -no-map runs once per prefix")
    ap.add_argument("--repeat-lines", action="store_true", help="the repeated runs write mapping lines (-tophits -queryids) instead of -no-map")
    ap.add_argument("--repeat-extra", default="", help="arguments appended to every repeated run (e.g. '-shard keys -key-shards 4 -batch-size 1000000')")
    ap.add_argument("--repeat-threads", default="", help="comma list: the repeated -no-map runs once per -threads value instead of pauses")
    ap.add_argument("--runs", default="mcq_nomap,mcq_map,mcq_tophits_ids", help="which of the three mcq runs (bench.py's e2e leg: mcq_nomap,mcq_tophits_ids)")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    build.build_library()
    import scale_util
    shm = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
    db = os.path.join(shm, f"mc_e2e_{os.getpid()}")
    fa, fa_small, o, oref = db + "_reads.fa", db + "_small.fa", db + "_o.txt", db + "_oref.txt"
    res = {"scale": args.scale, "reads": args.reads, "host_cpus_granted": scale_util.effective_cpus()}
    try:
        c2 = dict(bench.CFG2); c2["genera"] = max(2, int(round(c2["genera"] * args.scale)))
        spec = synthdb.phylogeny(**c2)
        shards = max(1, int(np.ceil(spec.total_bases // 112 * 16 / 1.4e9)))
        t0 = time.time()
        gdb, info = synthdb.build_database(spec, shards=shards, max_candidates=2, write_to=db, report=lambda m: print(m, file=sys.stderr, flush=True))
        res["collection"] = {"targets": len(spec.targets), "bases": int(spec.total_bases), "locations": int(gdb.info()[7]),
                             "database_file_bytes": sum(os.path.getsize(db + e) for e in (".meta", ".cache0")), "build_and_write_s": round(time.time() - t0, 1)}
        gdb.close()
        gen = synthdb.GpuSynth(0)
        P = synthdb.read_params(spec, 3100)
        with open(fa, "wb"):
            pass
        done = 0
        while done < args.reads:
            m = min(2_000_000, args.reads - done)
            rows = torch.zeros((m, P.row_bytes), dtype=torch.uint8, device="cuda:0")
            gen.reads(spec, P, done, m, rows)
            host = rows.cpu().numpy()[:, :150]
            tmpf = fa + ".part"
            e2e_bench.write_fasta(tmpf, host)
            if done == 0:
                e2e_bench.write_fasta(fa_small, host[: args.cpu_reads])
            with open(fa, "ab") as dst, open(tmpf, "rb") as src:
                dst.write(src.read())
            os.remove(tmpf)
            done += m
        del rows
        torch.cuda.empty_cache()
        res["fasta_bytes"] = os.path.getsize(fa)
        mcq = build.MCQ
        env = dict(os.environ, MCQ_PROFILE="1")
        for name, extra in (("mcq_nomap", ["-no-map"]), ("mcq_map", []), ("mcq_tophits_ids", ["-tophits", "-queryids"])):
            if name not in args.runs.split(","):
                continue
            wall, prof = timed([mcq, "query", db, fa] + extra + ["-out", o], env=env)
            q, ms = e2e_bench.speed_of(o)
            res[name] = {"wall_s": round(wall, 2), "query_ms": ms, "database_load_and_startup_s": round(wall - ms / 1e3, 2),
                         "Mreads_per_min_query_phase": round(q / (ms / 1e3) * 60 / 1e6, 1), "Mreads_per_min_wall": round(q / wall * 60 / 1e6, 1), "profile": prof}
            print(name, res[name], flush=True)
        for bs in filter(None, args.batch_sizes.split(",")):
            for name, extra in (("mcq_nomap", ["-no-map"]), ("mcq_map", []), ("mcq_tophits_ids", ["-tophits", "-queryids"])):
                wall, prof = timed([mcq, "query", db, fa] + extra + ["-batch-size", bs, "-out", o], env=env)
                q, ms = e2e_bench.speed_of(o)
                res[f"{name}_batch_{bs}"] = {"wall_s": round(wall, 2), "query_ms": ms, "profile": prof}
                print(name, bs, res[f"{name}_batch_{bs}"], flush=True)
        for np_ in filter(None, args.pipes.split(",")):
            for name, extra in (("mcq_nomap", ["-no-map"]), ("mcq_tophits_ids", ["-tophits", "-queryids"])):
                wall, prof = timed([mcq, "query", db, fa] + extra + ["-out", o], env=dict(env, MC_PIPES=np_))
                q, ms = e2e_bench.speed_of(o)
                res[f"{name}_pipes_{np_}"] = {"wall_s": round(wall, 2), "query_ms": ms, "profile": prof}
                print(name, "pipes", np_, res[f"{name}_pipes_{np_}"], flush=True)
        if args.repeat_nomap:
            runs = []
            plan = [(0.0, [])] * args.repeat_nomap + [(args.sleep, [])] * args.repeat_nomap
            if args.repeat_threads:
                plan = [(0.0, ["-threads", t]) for t in args.repeat_threads.split(",") for _ in range(args.repeat_nomap)]
            prefixes = [p.split() for p in args.repeat_prefix.split(";")] if args.repeat_prefix else [[]]
            if args.repeat_prefix:
                plan = [(0.0, [])] * args.repeat_nomap
            for prefix in prefixes:
              for pause, extra in plan:
                time.sleep(pause)
                wall, prof = timed(prefix + [mcq, "query", db, fa] + (["-tophits", "-queryids"] if args.repeat_lines else ["-no-map"]) + extra + args.repeat_extra.split() + ["-out", o], env=env)
                q, ms = e2e_bench.speed_of(o)
                runs.append({"prefix": " ".join(prefix), "slept_s": pause, "args": extra, "wall_s": round(wall, 2), "query_ms": ms, "profile": prof})
                print("mcq_nomap again", runs[-1], flush=True)
            res["mcq_nomap_repeats"] = runs
        if args.profile_nomap:
            import csv, glob, shutil
            runs = []
            for i in range(args.profile_nomap):
                pd = os.path.join(shm, f"mc_prof_{os.getpid()}_{i}")
                wall, prof = timed(["rocprofv3", "--kernel-trace", "--stats", "-d", pd, "-o", "p", "--output-format", "csv", "--", mcq, "query", db, fa, "-no-map", "-out", o],
                                   env=dict(env, TMPDIR="/tmp"))
                q, ms = e2e_bench.speed_of(o)
                kern = {}
                for fn in glob.glob(os.path.join(pd, "**", "*kernel_stats.csv"), recursive=True):
                    for row in csv.DictReader(open(fn)):
                        kern[row["Name"]] = kern.get(row["Name"], 0.0) + float(row["TotalDurationNs"]) / 1e6
                top = sorted(kern.items(), key=lambda kv: -kv[1])[:8]
                query_kernels = {k: round(v, 1) for k, v in top if "table_" not in k}
                runs.append({"wall_s": round(wall, 2), "query_ms": ms, "profile": prof, "kernel_total_ms_top": query_kernels})
                print("mcq_nomap profiled", runs[-1], flush=True)
                shutil.rmtree(pd, ignore_errors=True)
            res["mcq_nomap_profiled"] = runs
        if args.key_shards:
            ks = ["-shard", "keys", "-key-shards", str(args.key_shards), "-batch-size", "1000000"]
            wall, prof = timed([mcq, "query", db, fa, "-no-map"] + ks + ["-out", o], env=env)
            q, ms = e2e_bench.speed_of(o)
            res["mcq_key_shards_nomap"] = {"shards": args.key_shards, "wall_s": round(wall, 2), "query_ms": ms, "database_load_and_startup_s": round(wall - ms / 1e3, 2),
                                           "Mreads_per_min_query_phase": round(q / (ms / 1e3) * 60 / 1e6, 1)}
            print("key shards", res["mcq_key_shards_nomap"], flush=True)
            o2 = o + ".ks"
            timed([mcq, "query", db, fa_small, "-tophits", "-queryids", "-out", o])
            timed([mcq, "query", db, fa_small, "-tophits", "-queryids"] + ks + ["-out", o2])
            a = [l for l in open(o) if not l.startswith("#")]
            b = [l for l in open(o2) if not l.startswith("#")]
            os.remove(o2)
            res["key_shards_identical_mapping_lines"] = {"single_table": len(a), "key_shards": len(b), "differing": sum(x != y for x, y in zip(a, b)) + abs(len(a) - len(b))}
            print("key shards vs single table", res["key_shards_identical_mapping_lines"], flush=True)
        ref = os.path.join(ROOT, "oracle", "_ref", "metacache_u32")
        if os.path.exists(ref) and not args.no_ref:
            wall, _ = timed([ref, "query", db, fa, "-no-map", "-out", oref])
            q, ms = e2e_bench.speed_of(oref)
            res["reference_cpu_nomap"] = {"wall_s": round(wall, 2), "query_ms": ms, "database_load_and_startup_s": round(wall - ms / 1e3, 2),
                                          "Mreads_per_min_query_phase": round(q / (ms / 1e3) * 60 / 1e6, 2), "Mreads_per_min_wall": round(q / wall * 60 / 1e6, 2)}
            print("reference", res["reference_cpu_nomap"], flush=True)
            timed([ref, "query", db, fa_small, "-tophits", "-queryids", "-out", oref])
            timed([mcq, "query", db, fa_small, "-tophits", "-queryids", "-out", o])
            a = sorted(l for l in open(oref) if not l.startswith("#"))
            b = sorted(l for l in open(o) if not l.startswith("#"))
            res["identical_mapping_lines"] = {"reference": len(a), "mcq": len(b), "differing": sum(x != y for x, y in zip(a, b)) + abs(len(a) - len(b))}
    finally:
        for f in (db + ".meta", db + ".cache0", fa, fa_small, o, oref, fa + ".part"):
            if os.path.exists(f):
                os.remove(f)
    print(json.dumps(res, indent=1))
    if args.out:
        os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
