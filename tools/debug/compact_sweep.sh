# compact location store against the 8-byte one at full scale and at 45 Gbp
python -m pytest tests/test_gpu_scale.py tests/test_gpu_parity.py tests/test_gpu_mode_k.py -q -x 2>&1 | tail -3
for C in 1 0; do
  echo "=== MC_COMPACT_LOCATIONS=$C"
  MC_COMPACT_LOCATIONS=$C python tools/tune_big.py --scale 1 --big-min 256 --batch 5000000 2>&1 | grep big_min
  MC_COMPACT_LOCATIONS=$C python tools/tune_big.py --scale 0.3 --big-min 256 --batch 2000000 2>&1 | grep big_min
done
