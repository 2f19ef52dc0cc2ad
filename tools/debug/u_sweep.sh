set -e
for U in 32 48; do
  echo "=== MC_BIG_U=$U"
  MC_HIPCC_FLAGS="-DMC_BIG_U=$U" python -c "
from metacache_amd import build
build.build_library(force=True)" > /dev/null 2>&1
  python tools/tune_big.py --scale 1 --big-min 256 2>&1 | grep big_min
done
