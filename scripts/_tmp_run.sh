timeout 200 python -m pytest tests/test_gpu_parts.py tests/test_gpu_target_ranges.py tests/test_gpu_pipeline.py -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3
timeout 200 python tools/target_range_bench.py --scale 0.1 --whole-only --reads 2000000 2>&1 | grep -E '^\{"contexts"' | cut -c1-400
