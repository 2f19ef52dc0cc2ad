echo "=== two pipes"
python tools/tune_big.py --scale 1 --big-min 256 --batch 5000000 --two-pipes 2>&1 | grep "big_min\|two_pipes\|Error\|error" | cut -c1-700
echo "=== MC_LANE_FUSION=1"
MC_LANE_FUSION=1 python tools/tune_big.py --scale 1 --big-min 256 --batch 5000000 2>&1 | grep "big_min" | cut -c1-700
echo "=== load factor 0.3"
python tools/tune_big.py --scale 1 --big-min 256 --batch 5000000 --load-factor 0.3 2>&1 | grep "big_min" | cut -c1-700
echo "=== pytest -m gpu"
python -m pytest tests -m gpu -q -x 2>&1 | tail -5
