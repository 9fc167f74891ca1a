import torch, time
x = torch.empty(281*1024*1024//8, dtype=torch.float64, device='cuda')
y = torch.empty_like(x)
for name, fn in (("fill", lambda: x.zero_()), ("copy", lambda: y.copy_(x))):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): fn()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e)/10
    nbytes = x.numel()*8*(1 if name=="fill" else 2)
    print(name, "%.1f us"%(ms*1e3), "%.2f TB/s"%(nbytes/ms/1e9))
