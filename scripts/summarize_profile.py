#!/usr/bin/env python3
"""Condenses a gpurun_out/prof_<tag>/ directory (scripts/profile.sh) into profiles/<tag>_*.{csv,md}."""
import collections
import csv
import glob
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(src, "summary") if len(sys.argv) > 2 else os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)


def short(k):
    if "query_kernel<true>" in k: return "query_kernel<fused>"
    if "query_kernel<false>" in k: return "query_kernel<unfused>"
    if "sketch_probe_lane" in k: return "sketch_probe_lane"
    if "mid_cands_kernel" in k: return "mid_cands"
    if "hash_cands_kernel<9" in k: return "hash_cands_256"
    if "hash_cands_kernel<10" in k: return "hash_cands_512"
    if "hash_cands_kernel<11" in k: return "hash_cands_1024"
    if "hash_cands_kernel" in k: return "hash_cands"
    if "big_cands_kernel<10" in k: return "big_cands"
    if "big_cands_kernel<11" in k: return "big_cands_2"
    if "big_filter_kernel" in k:                                  # <WAVES, COMPACT, EPL, ...>: EPL = 1 is the first instance
        args = k.split("big_filter_kernel<", 1)[1].split(">")[0].split(",") if "big_filter_kernel<" in k else []
        return "big_filter" if len(args) < 3 or args[2].strip() in ("1u", "1") else "big_filter_2"
    if "gw_filter_stream_kernel" in k: return "big_filter_2"
    if "gw_filter_kernel" in k: return "big_filter"
    if "gw_count_kernel<9" in k: return "big_count"             # filtered lists up to 256 (most reads)
    if "gw_count_kernel<10" in k: return "big_count_512"
    if "gw_count_kernel<11" in k: return "big_count_2"
    if "gw_filter2_kernel" in k: return "big_filter_2rounds"
    if "gw_compact_kernel" in k: return "gw_compact"
    if "gw_sort" in k or "gw_sorted" in k: return "gw_sorted_cands"
    if "big_count_kernel<10" in k: return "big_count"
    if "big_count_kernel<11" in k: return "big_count_2"
    for n in ("sketch_lane", "probe_cands", "sort_candidates", "plan_kernel", "scan_block_sums", "scan_of_sums", "scan_apply", "batch_stats", "emit_pairs", "chunk_sketch", "chunk_probe", "chunk_finish", "flag_count", "table_seal", "build_sketch_lanes", "own_count", "own_emit", "union_copy"):
        if n in k: return n
    return None


# 1. rocprofv3 --kernel-trace --stats summary, our kernels only (torch's synthetic-data kernels dropped)
rows = list(csv.DictReader(open(os.path.join(src, "trace", "trace_kernel_stats.csv"))))
with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "StdDev"])
    for r in rows:
        if "mcamd" in r["Name"]:
            w.writerow([r["Name"].replace("(anonymous namespace)::", "").split("(")[0], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["MinNs"], r["MaxNs"], r["StdDev"]])

# 2. PMC counters per kernel (mean per dispatch over the dispatches of each separate --pmc pass)
res = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob(os.path.join(src, "pmc_*", "pmc_counter_collection.csv")):
    for r in csv.DictReader(open(fn)):
        s = short(r["Kernel_Name"])
        if s:
            res[s][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(os.path.join(dst, f"{tag}_pmc_summary.csv"), "w") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "counter", "dispatches", "mean_per_dispatch"])
    for k in sorted(res):
        for c in sorted(res[k]):
            v = res[k][c]
            w.writerow([k, c, len(v), f"{sum(v) / len(v):.6g}"])
print("wrote", dst)
