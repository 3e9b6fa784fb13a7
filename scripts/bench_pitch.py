"""Does a power-of-two row pitch of the sx8 operands slow the NT split GEMM (all rows of a tile slice in few
memory channels)?  Same GEMMs with operand rows padded by `pad` floats."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wireframe-3d-prediction_amd"))
import torch
from wf3d import ops
dev = torch.device("cuda:0")
M = 131072


def padded(t, pad):
    if pad == 0:
        return t
    buf = torch.empty(t.shape[0], t.shape[1] + pad, device=dev)
    v = buf[:, :t.shape[1]]
    v.copy_(t)
    return v


def timeit(fn, n=9):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return statistics.median(ts)


for K, N in [(512, 1024), (1024, 2048), (2048, 1024), (1024, 512)]:
    A, B = ops.split_rows(torch.randn(M, K, device=dev)), ops.split_rows(torch.randn(N, K, device=dev))
    out = torch.empty(M, N, device=dev)
    cfgs = [(0, 0), (0, 32), (32, 32), (0, 8), (8, 8), (0, 96)]
    ops_ = [(padded(A, pa), padded(B, pb)) for pa, pb in cfgs]
    times = [[] for _ in cfgs]
    for a, b in ops_:
        ops.gemm_split(a, b, out=out)
    torch.cuda.synchronize()
    for r in range(9):
        for i, (a, b) in enumerate(ops_):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); ops.gemm_split(a, b, out=out); e1.record(); torch.cuda.synchronize()
            times[i].append(e0.elapsed_time(e1))
    print(f"K={K} N={N}: " + "  ".join(f"A{pa:2d}/B{pb:2d}: {statistics.median(t):.3f}" for (pa, pb), t in zip(cfgs, times)))
