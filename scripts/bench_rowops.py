"""HBM-bound row kernels at the encoder's shapes (M = 32*4096 rows): GB/s achieved."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wireframe-3d-prediction_amd"))
import torch
from wf3d import ops
dev = torch.device("cuda:0")
M = 131072


def timeit(fn, n=11):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return statistics.median(ts)


for D in (512, 1024, 2048):
    z, dh = torch.randn(M, D, device=dev), torch.randn(M, D, device=dev)
    g, b = torch.ones(D, device=dev), torch.zeros(D, device=dev)
    mu, rs, h = ops.ln_prep(z, g, b, ops.ACT_RELU)
    dzs = torch.empty_like(z)
    t1 = timeit(lambda: ops.ln_prep(z, g, b, ops.ACT_RELU))
    t2 = timeit(lambda: ops.ln_act_bwd(dh, z, mu, rs, g, b, ops.ACT_RELU, dz_split=dzs, want_dz=False))
    print(f"D={D}: ln_prep {t1:7.1f} us = {M * D * 8 / t1 / 1e6:5.2f} TB/s   ln_act_bwd(sx8 out) {t2:7.1f} us = {M * D * 12 / t2 / 1e6:5.2f} TB/s")
