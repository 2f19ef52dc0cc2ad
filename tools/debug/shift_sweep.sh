for S in 3; do
  echo "=== MC_BIG_MIN_SHIFT=$S"
  MC_HIPCC_FLAGS="-DMC_BIG_MIN_SHIFT=$S" python -c "
from metacache_amd import build
build.build_library(force=True)" > /dev/null 2>&1
  python -m pytest tests/test_gpu_scale.py -q -x 2>&1 | tail -2
  python tools/tune_big.py --scale 1 --big-min 256 --batch 5000000 2>&1 | grep big_min
  python tools/tune_big.py --scale 0.3 --big-min 256 --batch 2000000 2>&1 | grep big_min
done
