for S in 0.1 0.03; do
python tools/tune_big.py --scale $S --big-min 256,128,64 --batch 1000000 --load-factor 0.3 2>&1 | grep big_min | cut -c1-500
done
