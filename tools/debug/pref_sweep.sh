run() {
  echo "=== $1 COUNT_BPC=$2"
  MC_HIPCC_FLAGS="$1" python -c "
from metacache_amd import build
build.build_library(force=True)" > /dev/null 2>&1
  MC_BIG_COUNT_BPC=$2 python tools/tune_big.py --scale 1 --big-min 256 --batch 5000000 --load-factor 0.3 2>&1 | grep big_min | cut -c1-600
}
run "-DMC_BIG_COUNT_PREFETCH=1 -DMC_BIG_COUNT_WPE=6" 6
run "-DMC_BIG_COUNT_PREFETCH=1 -DMC_BIG_COUNT_WPE=5" 5
run "-DMC_BIG_COUNT_PREFETCH=0 -DMC_BIG_COUNT_WPE=6" 6
