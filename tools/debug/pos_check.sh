python -m pytest tests/test_gpu_scale.py -q -x 2>&1 | grep "^E  \|passed\|failed" | head -8 | cut -c1-500
timeout 900 python tools/long_reads_scale.py --scale 1 --lengths 500,300 --out gpurun_out/long_reads_r02_full.json 2>&1 | grep read_len | cut -c1-700
