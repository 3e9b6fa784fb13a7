import torch, time
a = torch.randn(131072, 2048, device="cuda"); b = torch.empty_like(a)
for n in (2048, 1024):
    x, y = a[:, :n].contiguous(), torch.empty(131072, n, device="cuda")
    for _ in range(3): y.copy_(x)
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): y.copy_(x)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20
    print(f"torch copy_ 131072 x {n}: {t * 1e3:.1f} us  {2 * x.numel() * 4 / t / 1e6:.0f} GB/s")
    for _ in range(3): torch.add(x, 1.0, out=y)
    e0.record()
    for _ in range(20): torch.add(x, 1.0, out=y)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20
    print(f"torch add(x, 1, out=y) 131072 x {n}: {t * 1e3:.1f} us  {2 * x.numel() * 4 / t / 1e6:.0f} GB/s")
