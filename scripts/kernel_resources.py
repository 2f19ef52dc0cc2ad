#!/usr/bin/env python3
"""Registers / LDS / scratch of every kernel in libmetacache_amd.so (from the code objects' metadata notes):
  python scripts/kernel_resources.py [substring ...]"""
import glob, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "metacache_amd", "lib", "libmetacache_amd.so")
LLVM = "/opt/rocm/lib/llvm/bin"
with tempfile.TemporaryDirectory() as d:
    tmp = os.path.join(d, "lib.so")
    os.symlink(LIB, tmp)
    subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", tmp], cwd=d, stdout=subprocess.DEVNULL, check=True)
    for co in sorted(glob.glob(os.path.join(d, "*gfx950"))):
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
        for blk in notes.split("- .agpr_count:")[1:]:
            g = lambda k: (re.search(rf"\.{k}:\s*(\S+)", blk) or [None, "?"])[1]
            name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
            if len(sys.argv) > 1 and not any(s in name for s in sys.argv[1:]):
                continue
            print(f"vgpr {g('vgpr_count'):>4} sgpr {g('sgpr_count'):>4} lds {g('group_segment_fixed_size'):>6} scratch {g('private_segment_fixed_size'):>5} spill {g('vgpr_spill_count'):>3}  {name[:150]}")
