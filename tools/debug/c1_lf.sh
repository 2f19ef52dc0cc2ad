for LF in 0.3 0.5 0.8; do echo "=== configs[1] load factor $LF"; python bench.py --config 1 --load-factor $LF --steps 10 --warmup 2 --cpu-seconds 0 --gather-gib 0 2>/dev/null | python -c "
import sys, json
r = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print(json.dumps({'lf': r['config']['load_factor'], 'value': r['value'], 'ms_per_step': r['ms_per_step'], 'hbm_used_GB': r['config']['hbm_used_GB'], 'probe_cands_ms': r['roofline']['kernel_ms']['probe_cands'], 'sketch_lane_ms': r['roofline']['kernel_ms']['sketch_lane']}))"; done
echo "=== configs[1] with the reference CPU leg (thread sweep)"; python bench.py --config 1 --steps 10 --warmup 2 --gather-gib 8 2>/dev/null | tail -1 > gpurun_out/bench_r02_config1.json; tail -c 1500 gpurun_out/bench_r02_config1.json
