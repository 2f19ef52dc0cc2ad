#!/bin/bash
# Two PMC passes (instruction counts, wait / busy cycles) of a command on the GPU box: bash scripts/pmc_quick.sh <tag> <command...>
set -u
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM_RD" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  (cd $REPO && rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$name -o pmc -- "$@") > $OUT/pmc_$name.log 2>&1
done
mkdir -p $OUT/trace; echo "Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs,StdDev" > $OUT/trace/trace_kernel_stats.csv
python $REPO/scripts/summarize_profile.py $TAG $OUT
find $OUT -name "pmc_counter_collection.csv" -size +20M -delete
grep -E "^\"?(gw_filter|gw_count|probe_cands|sketch_lane_kernel)" $OUT/summary/${TAG}_pmc_summary.csv
