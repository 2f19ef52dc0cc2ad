# compact store, full scale: blocks per CU of big_filter_kernel (26 KB of LDS per block) and of the second counting instance
python -c "
from metacache_amd import build
build.build_library(force=True)" > /dev/null 2>&1
python -m pytest tests/test_gpu_scale.py tests/test_gpu_parity.py tests/test_gpu_mode_k.py -q -x 2>&1 | tail -3
for F in 5 6; do for C in 4 2; do
  echo "=== FILTER_BPC=$F COUNT2_BPC=$C"
  MC_BIG_FILTER_BPC=$F MC_BIG_COUNT2_BPC=$C python tools/tune_big.py --scale 1 --big-min 256 --batch 5000000 2>&1 | grep big_min
done; done
MC_BIG_FILTER_BPC=6 python tools/tune_big.py --scale 0.3 --big-min 256 --batch 2000000 2>&1 | grep big_min
MC_COMPACT_LOCATIONS=0 python tools/tune_big.py --scale 1 --big-min 256 --batch 5000000 2>&1 | grep big_min
