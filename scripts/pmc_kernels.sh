#!/bin/bash
# Kernel trace + two PMC passes (separate runs) of bench.py --long-reads, counters for the kernels matching the regular expression $1 only:
#   gpurun -- bash scripts/pmc_kernels.sh "gw_filter_stream|gw_sorted_cands"     -> gpurun_out/pmc_block/summary.txt
set -u
PAT=${1:-gw_count_block}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_block
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--long-reads --steps 3 --warmup 1 --cpu-seconds 0 --repeats 1 --no-pipeline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python $REPO/bench.py $ARGS > $OUT/trace.log 2>&1
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-30)
  rocprofv3 --pmc $grp --kernel-include-regex "$PAT" --output-format csv -d $OUT/pmc_$name -o pmc -- python $REPO/bench.py $ARGS > $OUT/pmc_$name.log 2>&1
done
find $OUT -name "*_kernel_trace.csv" -delete
python - <<PY
import csv, glob, collections
res = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob("$OUT/pmc_*/pmc_counter_collection.csv"):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"].split("(")[0][-60:]
        res[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$OUT/summary.txt", "w") as f:
    for k in sorted(res):
        for c in sorted(res[k]):
            v = res[k][c]
            f.write(f"{k} {c} {len(v)} {sum(v)/len(v):.6g}\n")
for r in csv.DictReader(open(glob.glob("$OUT/trace/*kernel_stats.csv")[0])):
    if "gw_" in r["Name"] or "segmented" in r["Name"]:
        print(r["Calls"], float(r["AverageNs"]) / 1e6, r["Name"][:100])
PY
find $OUT -name "pmc_counter_collection.csv" -delete
cat $OUT/summary.txt
