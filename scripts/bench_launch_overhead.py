"""CPU cost per op launch through the ctypes binding (no synchronisation inside the loop)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wireframe-3d-prediction_amd"))
import torch
from wf3d import ops
dev = torch.device("cuda:0")
z = torch.randn(32, 512, device=dev)
W = torch.randn(256, 512, device=dev)
for name, fn in [("row_stats", lambda: ops.row_stats(z)), ("gemm 32x256x512", lambda: ops.gemm(z, W, ops.NT)),
                 ("torch.empty", lambda: torch.empty(32, 512, device=dev)), ("colsum", lambda: ops.colsum(z))]:
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2000):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"{name:20s} {(t1 - t0) / 2000 * 1e6:6.1f} us of CPU per call")
