#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_fused
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/tools/tune_gw.py --scale 1 --set lookup_fusion=1,0 --pipes 1 --steps 3"
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-30)
  rocprofv3 --pmc $grp --kernel-include-regex "gw_lookup_filter_count|gw_filter_count|sketch_probe_lane" --output-format csv -d $OUT/pmc_$name -o pmc -- $CMD > $OUT/pmc_$name.log 2>&1
done
python - <<PY
import csv, glob, collections
res = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob("$OUT/pmc_*/pmc_counter_collection.csv"):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"].split("(")[0][-50:]
        res[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$OUT/summary.txt", "w") as f:
    for k in sorted(res):
        for c in sorted(res[k]):
            v = res[k][c]
            v = [x for x in v if x > 0.01 * max(v)]     # (launches with work)
            f.write(f"{k} {c} {len(v)} {sum(v)/max(len(v),1):.6g}\n")
PY
find $OUT -name "pmc_counter_collection.csv" -delete
cat $OUT/summary.txt
