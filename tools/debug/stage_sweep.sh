python -m pytest tests/test_gpu_scale.py -q -x 2>&1 | tail -3
echo "=== stage 1536 (default build)"; python tools/tune_big.py --scale 1 --big-min 256 --batch 5000000 2>&1 | grep big_min
for S in 2048 1024; do
  echo "=== MC_BIG_STAGE=$S"
  MC_HIPCC_FLAGS="-DMC_BIG_STAGE=$S" python -c "
from metacache_amd import build
build.build_library(force=True)" > /dev/null 2>&1
  python tools/tune_big.py --scale 1 --big-min 256 --batch 5000000 2>&1 | grep big_min
done
