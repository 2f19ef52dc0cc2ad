for F in 7; do
  echo "=== FILTER_BPC=$F"
  MC_BIG_FILTER_BPC=$F python tools/tune_big.py --scale 1 --big-min 256 --batch 5000000 --load-factor 0.3 2>&1 | grep big_min | cut -c1-600
done
echo "=== T1=15 BPC=6"
MC_HIPCC_FLAGS="-DMC_BIG_T1=15 -DMC_BIG_T2=13" python -c "
from metacache_amd import build
build.build_library(force=True)" > /dev/null 2>&1
MC_BIG_FILTER_BPC=6 python tools/tune_big.py --scale 1 --big-min 256 --batch 5000000 --load-factor 0.3 2>&1 | grep big_min | cut -c1-600
