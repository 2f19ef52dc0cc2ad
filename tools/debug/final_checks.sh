python -m pytest tests -m gpu -q -x 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 1200 python tools/long_reads_scale.py --scale 1 --lengths 250,500 --out gpurun_out/long_reads_r02_full.json 2>&1 | grep read_len | cut -c1-700
