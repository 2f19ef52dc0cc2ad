python -m pytest tests/test_gpu_scale.py tests/test_gpu_parity.py tests/test_gpu_mode_k.py -q -x 2>&1 | tail -4
python tools/tune_big.py --scale 1 --big-min 256 --batch 5000000 --load-factor 0.3 2>&1 | grep big_min | cut -c1-600
timeout 900 python tools/long_reads_scale.py --scale 0.3 --lengths 300,500 --out gpurun_out/long_reads_r02b.json 2>&1 | grep read_len | cut -c1-700
