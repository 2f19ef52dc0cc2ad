python tools/ref_parity_midscale.py --scale 0.05 --reads 300000 --out gpurun_out/ref_parity_midscale.json 2>&1 | tail -12
echo "--- modes at scale 0.1, one rank"
for M in R P K; do MASTER_ADDR=127.0.0.1 MASTER_PORT=29544 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 python bench.py --scale 0.1 --batch 1000000 --steps 6 --warmup 2 --gather-gib 0 --parity-reads 20000 --cpu-seconds 3 --mode $M --force-dist 2>/dev/null | python -c "
import sys, json
r = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(r['config']['mode'], r['value'], r['ms_per_step'], r['parity'], {k:v for k,v in r['roofline']['kernel_ms'].items() if v>0.05})"; done
