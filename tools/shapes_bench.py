"""Hot-path throughput for the read shapes of BASELINE configs[3] / configs[4] on ONE GPU (the 8-GPU runs are the driver's):
2 x 150 bp pairs (fragment 300-500, mate 2 reverse-complemented, maxWindowsInRange = 4) and long reads (log-normal lengths,
median 480, 200..19000 bp, 7.5 % substitutions, maxWindowsInRange = 2 + len/112), reads resident in HBM, against the
configs[1] database.  Prints one JSON object; not bench.py's `value`.

    python tools/shapes_bench.py [--pairs 1000000] [--long 200000]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from metacache_amd import api  # noqa: E402

LUT = None


def mutate(code, gen, rate):
    sub = torch.rand(code.shape, generator=gen, device=code.device) < rate
    shift = torch.randint(1, 4, code.shape, generator=gen, device=code.device, dtype=torch.uint8)
    return torch.where(sub, (code + shift) % 4, code)


def ragged_reads(gcode, G, GL, lengths, gen, rate):
    """reads of the given lengths from uniform positions, random strand; -> (flat ASCII uint8 with every read 4-byte aligned + 16
    slack bytes, offsets int64 [n])"""
    dev = gcode.device
    n = lengths.numel()
    gi = torch.randint(0, G, (n,), generator=gen, device=dev)
    st = (torch.rand(n, generator=gen, device=dev) * (GL - lengths).clamp(min=1)).long()
    pad = (lengths + 3) // 4 * 4
    offs = torch.cumsum(pad, 0) - pad
    total = int(pad.sum())
    rid = torch.repeat_interleave(torch.arange(n, device=dev), lengths)
    pos = torch.arange(int(lengths.sum()), device=dev) - torch.repeat_interleave(torch.cumsum(lengths, 0) - lengths, lengths)
    flip = (torch.rand(n, generator=gen, device=dev) < 0.5)[rid]
    src = gi[rid] * GL + st[rid] + torch.where(flip, lengths[rid] - 1 - pos, pos)
    code = gcode[src]
    code = torch.where(flip, 3 - code, code)
    code = mutate(code, gen, rate)
    out = torch.zeros(total + 16, dtype=torch.uint8, device=dev)
    out[offs[rid] + pos] = LUT[code.long()]
    return out, offs


def run(db, seq, qinfo, max_win, n, nchars, K, steps):
    dev = seq.device
    out = torch.zeros((n, K, 4), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()              # the library works on its own stream: inputs prepared by torch must be complete

    def step():
        r = db.query_device(seq.data_ptr(), qinfo.data_ptr(), n, nchars, max_win_ptr=max_win.data_ptr())
        db.copy_results(out.data_ptr(), r.cands, n * K * 16)
        db.synchronize()
    step(); step()
    db.timing(True); db.timing_reset()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    el = (time.perf_counter() - t0) / steps
    db.timing(False)
    names = ("plan", "sketch_lane", "chunk_sketch", "chunk_probe", "probe_cands", "mid_cands_64", "mid_cands_128", "mid_cands_256", "hash_cands_256", "hash_cands_512", "hash_cands_1024", "query_wave", "scan", "sort_candidates")
    kt = {k: db.timing_get(k) for k in names}
    st = db.last_batch_stats()
    c = out.cpu().numpy().view(np.uint32)
    return el, {k: round(v[0] / max(v[1], 1), 4) for k, v in kt.items()}, st, float((c[:, 0, 1] > 0).mean())


def main():
    global LUT
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=1_000_000)
    ap.add_argument("--long", type=int, default=200_000)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    LUT = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    G, GL, K = 16, 5_000_000, 2
    genomes = bench.make_genomes(G, GL, seed=16)
    bld = api.Builder(target_id_bytes=2, max_candidates=K)
    for i, g in enumerate(genomes):
        bld.add_target(g, f"SYN_{i:06d}.1", parent_taxid=1000 + i)
    db = bld.finish(load=True)
    bld.free()
    gasc = torch.from_numpy(np.concatenate(genomes)).to(dev)
    gcode = torch.zeros_like(gasc)
    gcode[gasc == ord("C")] = 1; gcode[gasc == ord("G")] = 2; gcode[gasc == ord("T")] = 3
    del gasc
    gen = torch.Generator(device=dev); gen.manual_seed(4100)
    stride = db.stride
    res = {}

    # ---- pairs ---------------------------------------------------------------------------------------------------
    n = args.pairs
    frag = torch.randint(300, 501, (n,), generator=gen, device=dev)
    fseq, foff = ragged_reads(gcode, G, GL, frag, gen, 0.0)                       # error-free fragments, then two noisy mates
    idx = torch.arange(150, device=dev)[None, :]
    f1 = fseq[(foff[:, None] + idx)]
    tail = fseq[(foff + frag)[:, None] - 1 - idx]                                 # last 150 bases, reversed
    comp = torch.zeros(256, dtype=torch.uint8, device=dev)
    for a, b in zip(b"ACGT", b"TGCA"):
        comp[a] = b
    f2 = comp[tail.long()]
    seq = torch.zeros(n * 304 + 16, dtype=torch.uint8, device=dev)
    body = seq[: n * 304].view(n, 304)
    body[:, :150] = f1; body[:, 152:302] = f2
    noise = torch.rand((n, 304), generator=gen, device=dev) < 0.01
    valid = (body != 0) & noise
    body[valid] = ord("N")                                                       # 1 % ambiguous instead of substitutions: same k-mer loss
    qinfo = torch.zeros((n, 4), dtype=torch.int32, device=dev)
    qinfo[:, 0] = torch.arange(n, device=dev, dtype=torch.int32) * 304; qinfo[:, 1] = 150
    qinfo[:, 2] = qinfo[:, 0] + 152; qinfo[:, 3] = 150
    mw = torch.full((n,), 2 + 300 // stride, dtype=torch.int32, device=dev)
    el, kms, st, frac = run(db, seq, qinfo, mw, n, n * 304, K, args.steps)
    res["pairs_2x150"] = {"pairs_per_step": n, "ms_per_step": round(el * 1e3, 3), "Mreads_per_min": round(2 * n / el * 60 / 1e6, 1),
                          "kernel_ms": kms, "locations_per_pair": round(st["locations"] / n, 2), "pairs_with_candidate": frac}
    del seq, body, fseq, f1, f2, tail

    # ---- long reads ----------------------------------------------------------------------------------------------
    n = args.long
    ln = torch.exp(torch.randn(n, generator=gen, device=dev) * 0.9 + float(np.log(480.0))).clamp(200, 19000).long()
    seq, offs = ragged_reads(gcode, G, GL, ln, gen, 0.075)
    qinfo = torch.zeros((n, 4), dtype=torch.int32, device=dev)
    qinfo[:, 0] = offs.int(); qinfo[:, 1] = ln.int(); qinfo[:, 2] = offs.int()
    mw = (2 + ln // stride).int()
    el, kms, st, frac = run(db, seq, qinfo, mw, n, seq.numel() - 16, K, args.steps)
    bases = int(ln.sum())
    res["long_reads"] = {"reads_per_step": n, "bases_per_step": bases, "median_len": int(ln.median()), "max_len": int(ln.max()),
                         "ms_per_step": round(el * 1e3, 3), "Mreads_per_min": round(n / el * 60 / 1e6, 1),
                         "Gbases_per_s": round(bases / el / 1e9, 2), "kernel_ms": kms,
                         "locations_per_read": round(st["locations"] / n, 2), "reads_with_candidate": frac}
    print(json.dumps(res, indent=1))
    if args.out:
        os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(res, f, indent=1)
    db.close()


if __name__ == "__main__":
    main()
