MC_HIPCC_FLAGS="-DMC_BIG_COUNT2_WPE=3" python -c "
from metacache_amd import build
build.build_library(force=True)" > /dev/null 2>&1
for B in 6 4; do
echo "=== COUNT2 WPE=3 BPC=$B"
MC_BIG_COUNT2_BPC=$B python bench.py --pairs --batch 2500000 --steps 6 --warmup 2 --gather-gib 0 --cpu-seconds 0 2>/dev/null | python -c "
import sys, json
r = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(r['value'], r['ms_per_step'], {k:v for k,v in r['roofline']['kernel_ms'].items() if v>0.3})"
done
