for T in "16 13" "17 14"; do set -- $T
echo "=== POS T1=$1 T2=$2"
MC_HIPCC_FLAGS="-DMC_BIG_POS_T1=$1 -DMC_BIG_POS_T2=$2" python -c "
from metacache_amd import build
build.build_library(force=True)" > /dev/null 2>&1
timeout 600 python tools/long_reads_scale.py --scale 1 --lengths 500 --check 60 2>&1 | grep read_len | cut -c1-600
done
