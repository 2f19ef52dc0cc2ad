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
    """the kernel's own name as rocprofv3 prints it, without return type, namespaces and argument list: gw_filter_count_kernel<4u, 14u, false, 4u>
    (None for kernels that are not this library's)"""
    if "mcamd::" not in k:
        return None
    k = k.replace("(anonymous namespace)::", "")
    k = k.split("mcamd::", 1)[1]
    depth = 0
    for i, ch in enumerate(k):                                    # cut at the '(' of the argument list (template arguments may hold parentheses)
        if ch == "<": depth += 1
        elif ch == ">": depth -= 1
        elif ch == "(" and depth == 0:
            k = k[:i]
            break
    return k.strip()


def full(k):
    """the name as rocprofv3 prints it, whole (template arguments and all -- they may hold commas and parentheses), without the argument list"""
    k = k.replace("(anonymous namespace)::", "")
    depth = 0
    for i, ch in enumerate(k):
        if ch == "<": depth += 1
        elif ch == ">": depth -= 1
        elif ch == "(" and depth == 0:
            return k[:i].strip()
    return k.strip()


# 1. rocprofv3 --kernel-trace --stats summary, our kernels only (torch's synthetic-data kernels dropped)
rows = list(csv.DictReader(open(os.path.join(src, "trace", "trace_kernel_stats.csv"))))
with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "StdDev"])
    for r in rows:
        if "mcamd" in r["Name"]:
            w.writerow([full(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["MinNs"], r["MaxNs"], r["StdDev"]])

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

# 3. which table the passes saw: the bench line of the trace pass (every pass runs the same command) -- bench.py refuses a request count
# that was taken on another layout (pmc_layout_matches)
import json
import subprocess
line = None
log = os.path.join(src, "trace.log")
if os.path.exists(log):
    for l in open(log, errors="replace"):
        if l.startswith('{"metric"'):
            line = json.loads(l)
if line:
    rf = line["roofline"]
    try:
        head = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True, stderr=subprocess.DEVNULL).strip()
    except Exception:
        head = None
    with open(os.path.join(dst, f"{tag}_layout.json"), "w") as f:
        json.dump({"table_location_bytes": rf["table_location_bytes"], "table_list_align": rf["table_list_align"], "table_list_entries": rf["table_list_entries"], "table_direct_index": rf.get("table_direct_index", False),
                   "workload": line["config"]["workload"], "reads_per_step_per_gpu": line["config"]["reads_per_step_per_gpu"], "pairs": line["config"]["pairs"],
                   "ms_per_step_under_the_tracer": line["ms_per_step"], "kernel_ms_under_the_tracer": rf["kernel_ms"], "head": head,
                   "command": "scripts/profile.sh " + tag + " (bench.py --steps 4 --warmup 2 --cpu-seconds 0 --repeats 1 ...)"}, f, indent=1)
print("wrote", dst)
