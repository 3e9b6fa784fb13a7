"""Layer 0 of the per-point MLP at the cfg2 shape: fused kernels vs the GEMM + normalisation passes they replace."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wireframe-3d-prediction_amd"))
import torch
from wf3d import ops
dev = torch.device("cuda:0")
R, K, D = 131072, 8, 512
torch.manual_seed(0)
x, W, b = torch.randn(R, K, device=dev), torch.randn(D, K, device=dev) * 0.3, torch.randn(D, device=dev) * 0.1
g, be = torch.ones(D, device=dev), torch.zeros(D, device=dev)
dh = torch.randn(R, D, device=dev)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return statistics.median(ts)


z, mu, rs, hs = ops.first_layer_fwd(x, W, b, g, be, ops.ACT_RELU)
print(f"fused forward            {timeit(lambda: ops.first_layer_fwd(x, W, b, g, be, ops.ACT_RELU)):7.1f} us")
print(f"gemm + ln_prep           {timeit(lambda: ops.ln_prep(ops.gemm(x, W, ops.NT, bias=b), g, be, ops.ACT_RELU)):7.1f} us")
print(f"fused backward           {timeit(lambda: ops.ln_act_bwd_first(dh, z, x, mu, rs, g, be, ops.ACT_RELU)):7.1f} us")
def old_bwd():
    dz, a, c, d = ops.ln_act_bwd(dh, z, mu, rs, g, be, ops.ACT_RELU)
    return ops.gemm(dz, x, ops.TN)
print(f"ln_act_bwd + TN gemm     {timeit(old_bwd):7.1f} us")
