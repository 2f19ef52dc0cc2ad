#!/bin/bash
# FETCH_SIZE / TCC_EA0_RDREQ of rocprofv3 on the random-access shapes of the path (tools/gather_width.hip), for profiles/r05_fetch_calibration.md:
#   gpurun -- bash scripts/fetch_calibration.sh        -> gpurun_out/fetch_cal/{run.txt,pmc_*.txt}
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/fetch_cal
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 -o /tmp/gather_width $REPO/tools/gather_width.hip || exit 1
/tmp/gather_width 65536 > $OUT/run.txt 2>&1
for grp in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "FETCH_SIZE"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-30)
  rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$name -o pmc -- /tmp/gather_width 65536 > $OUT/pmc_$name.log 2>&1
  python $REPO/tools/pmc_sum.py $(find $OUT/pmc_$name -name "*counter_collection.csv" | head -1) > $OUT/pmc_$name.txt
done
find $OUT -name "*counter_collection.csv" -delete
cat $OUT/run.txt; cat $OUT/pmc_*.txt
