import sys,json
for l in sys.stdin:
    try: r=json.loads(l)
    except Exception: continue
    print({k:r[k] for k in r if k not in ("stats","kernel_ms","Mreads_per_min")}, {k:v for k,v in r["kernel_ms"].items() if v>0.3})
