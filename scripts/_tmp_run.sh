timeout 1200 python -m pytest tests/test_gpu_slots.py tests/test_gpu_parity.py tests/test_gpu_pipeline.py tests/test_cli_gpu.py -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3
for p in 8 8 5 5 6; do
echo "MC_PIPES=$p"
MC_PIPES=$p timeout 900 python tools/slot_path_bench.py --scale 1 --threads 32,16,8,32 --batch 4096 --seconds 2.5 2>&1 | grep -E "^\{'slots_united|Error|error" | cut -c1-400 | sed "s/'reads_with_other_candidates': 0, 'errors': \[\]//; s/'slots_united': True, //"
done
