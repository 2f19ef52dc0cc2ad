timeout 1200 python tools/slot_path_bench.py --scale 1 --threads 1,4,8,16,32,64 --batch 4096,16384,65536 --out gpurun_out/r06_slot_path.json 2>&1 | grep -E "^\{'slots_united|Error|error" | cut -c1-300 | sed "s/'reads_with_other_candidates': 0, 'errors': \[\]//"
timeout 1500 python bench.py > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err; tail -c 600 gpurun_out/r06_bench_default.err | tail -3
python - <<'PY'
import json
for l in open('gpurun_out/r06_bench_default.json'):
    if l.startswith('{"metric"'):
        d=json.loads(l); print(d['ms_per_step'], d['value'], d.get('value_range'), d['roofline']['frac'], d['roofline']['step_frac'], d['parity'], d['host_fed']['ms_per_step'], d.get('e2e'))
PY
