for B in 5 4 3; do echo "=== MC_BIG_FILTER_BPC=$B"; MC_BIG_FILTER_BPC=$B python tools/tune_big.py --scale 1 --big-min 256 --batch 5000000 2>&1 | grep big_min; done
