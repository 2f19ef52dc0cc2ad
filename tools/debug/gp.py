import ctypes as C, sys, json
sys.path.insert(0, "/root/repo")
import torch
from metacache_amd import build
lib = C.CDLL(build.build_gather_peak())
lib.mcg_gather_peak.argtypes = [C.c_uint64, C.POINTER(C.c_double)]
torch.cuda.init(); torch.zeros(1, device="cuda")
for gib in (0.25, 1, 8, 32, 96, 180):
    r = (C.c_double * 3)()
    rc = lib.mcg_gather_peak(int(gib * (1 << 30)), r)
    print(json.dumps({"GiB": gib, "rc": rc, "lane_private_64B_Greq_s": round(r[0] / 1e9, 2), "quad_64B_Greq_s": round(r[1] / 1e9, 2),
                      "wave_512B_Greq_s": round(r[2] / 1e9, 2), "wave_512B_loads_G_s": round(r[2] / 8.875 / 1e9, 2), "wave_512B_TB_s": round(r[2] * 64 / 1e12, 2)}), flush=True)
