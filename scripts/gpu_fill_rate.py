"""Raw HBM fill / copy rate at the sizes the condensing kernels write (64 models: 281 MB, 512 models: 2.25 GB): the ceiling of a
write-bound kernel.  python scripts/gpu_fill_rate.py [MB ...]"""
import sys
import torch
sizes = [int(a) for a in sys.argv[1:]] or [281, 2250]
for mb in sizes:
    x = torch.empty(mb * 1024 * 1024 // 8, dtype=torch.float64, device='cuda')
    y = torch.empty_like(x)
    for name, fn in (("fill", lambda: x.zero_()), ("copy", lambda: y.copy_(x))):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): fn()
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 10
        nbytes = x.numel() * 8 * (1 if name == "fill" else 2)
        print("%5d MB %s %.1f us %.2f TB/s" % (mb, name, ms * 1e3, nbytes / ms / 1e9))
    del x, y
