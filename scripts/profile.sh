#!/bin/bash
# Profiles bench.py on the GPU box (run through gpurun).  Usage: bash scripts/profile.sh <tag> [bench args...]
# Kernel trace and every PMC group are separate runs (rocprofv3 --pmc is never combined with tracing).
set -u
TAG=${1:-r02}; shift || true
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 4 --warmup 2 --cpu-seconds 0 --repeats 1 --host-fed 0 --gather-gib 0 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python $REPO/bench.py $ARGS > $OUT/trace.log 2>&1
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM_RD" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$name -o pmc -- python $REPO/bench.py $ARGS > $OUT/pmc_$name.log 2>&1
done
# the raw traces are large: keep the per-kernel aggregates only
python $REPO/scripts/summarize_profile.py $TAG $OUT
find $OUT -name "*_kernel_trace.csv" -delete; find $OUT -name "pmc_counter_collection.csv" -delete; find $OUT -name "*.log" -size +1M -delete
du -sh $OUT
