#!/usr/bin/env python3
"""Per-launch time of wf3d_gemm at the edge head's per-vertex shapes (M = sum of vertex counts) against the reduction
length: separates the fixed cost of a launch (first load, epilogue, drain) from the cost per 32-wide k slice, for the
fp32 MFMA loop and the bf16x3 (x3) loop.  `python scripts/bench_gemm_mid.py`"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "wireframe-3d-prediction_amd"))
from wf3d import ops  # noqa: E402


def timed(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    dev = torch.device("cuda:0")
    print(f"{'layout':6} {'M':>6} {'N':>5} {'K':>5}  fp32 us   x3 us")
    for layout, name in ((ops.NT, "NT"), (ops.NN, "NN"), (ops.TN, "TN")):
        for (M, N) in ((2048, 512), (2048, 1536), (8192, 512)):
            for K in (32, 128, 512, 2048):
                if layout == ops.NT:
                    a, b = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
                elif layout == ops.NN:
                    a, b = torch.randn(M, K, device=dev), torch.randn(K, N, device=dev)
                else:
                    a, b = torch.randn(K, M, device=dev), torch.randn(K, N, device=dev)
                out = torch.empty(M, N, device=dev)
                t0 = timed(lambda: ops.gemm(a, b, layout, out=out))
                t1 = timed(lambda: ops.gemm(a, b, layout, out=out, x3=True))
                print(f"{name:6} {M:6d} {N:5d} {K:5d}  {t0:7.1f}  {t1:7.1f}")


def wgrads():
    """dW[M, N] = dz^T a over K = sum-of-vertex-count rows (TN): small outputs, split-K + combine launch included."""
    dev = torch.device("cuda:0")
    print(f"{'wgrad':6} {'M':>6} {'N':>5} {'K':>5}  fp32 us   x3 us")
    for K in (2048, 8192):
        for (M, N) in ((512, 512), (1536, 512), (512, 256), (512, 3)):
            a, b = torch.randn(K, M, device=dev), torch.randn(K, N, device=dev)
            out = torch.empty(M, N, device=dev)
            t0 = timed(lambda: ops.gemm(a, b, ops.TN, out=out))
            t1 = timed(lambda: ops.gemm(a, b, ops.TN, out=out, x3=True))
            print(f"{'TN':6} {M:6d} {N:5d} {K:5d}  {t0:7.1f}  {t1:7.1f}")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "wgrad":
        wgrads()
    else:
        main()
