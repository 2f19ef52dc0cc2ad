#!/bin/bash
# kernel trace of bench.py --long-reads (one batch at a time): gpurun -- bash scripts/trace_long.sh  -> gpurun_out/trace_long/stats.txt
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/trace_long
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python $REPO/bench.py --long-reads --steps 4 --warmup 1 --cpu-seconds 0 --repeats 1 --no-pipeline --calibrate-scale 0 --selfcheck-seconds 0 --gather-gib 0 > $OUT/trace.log 2>&1
python - <<PY
import csv, glob
rows = list(csv.DictReader(open(glob.glob("$OUT/trace/*kernel_stats.csv")[0])))
with open("$OUT/stats.txt", "w") as f:
    for r in rows:
        if "mcamd" in r["Name"] and "synth" not in r["Name"]:
            f.write(f'{int(r["Calls"]):6d} {float(r["AverageNs"])/1e6:9.4f} ms  {float(r["TotalDurationNs"])/1e6:9.2f} ms  {r["Name"][:110]}\n')
PY
find $OUT -name "*_kernel_trace.csv" -delete
sort -k4 -n -r $OUT/stats.txt | head -40
