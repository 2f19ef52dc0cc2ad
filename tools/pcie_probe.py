"""H2D / D2H bandwidth of the box (pinned host memory), for the end-to-end discussion in DESIGN.md §5."""
import time
import torch
dev = torch.device("cuda", 0)
for mb in (16, 256, 1024):
    h = torch.empty(mb << 20, dtype=torch.uint8).pin_memory()
    d = torch.empty(mb << 20, dtype=torch.uint8, device=dev)
    for name, (src, dst) in (("H2D", (h, d)), ("D2H", (d, h))):
        dst.copy_(src, non_blocking=True); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        print(f"{name} {mb} MiB: {mb / 1024 / dt:.1f} GiB/s")
