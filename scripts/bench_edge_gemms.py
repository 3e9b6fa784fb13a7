#!/usr/bin/env python3
"""The edge MLP's four split GEMMs (forward 512->256->128, dgrad back) at cfg2's 64,512 and cfg5's 1,044,480 edge rows:
time per launch under the default kernel choice; run with WF3D_SPLIT_DMA=3 / 6 to force the 256x128 / 256x256 kernel."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "wireframe-3d-prediction_amd"))
from wf3d import ops  # noqa: E402


def timed(fn, reps=20):
    for _ in range(3):
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
    for M in (64512, 1044480):
        for (N, K) in ((256, 512), (128, 256), (256, 128), (512, 256)):
            a = ops.split_rows(torch.randn(M, K, device=dev))
            b = ops.split_rows(torch.randn(N, K, device=dev))
            out = torch.empty(M, N, device=dev)
            t = timed(lambda: ops.gemm_split(a, b, out=out))
            print(f"M={M:8d} N={N:4d} K={K:4d}  {t:8.1f} us  {2.0 * M * N * K / t / 1e6:6.1f} TF")


def wgrads():
    """the edge MLP's two weight gradients, dW[256, 512] and dW^T[256, 128], over the edge rows (TN on sx8 operands)"""
    dev = torch.device("cuda:0")
    for K in (64512, 1044480):
        for (Mo, No) in ((256, 512), (256, 128)):
            a = ops.split_rows(torch.randn(K, Mo, device=dev))
            b = ops.split_rows(torch.randn(K, No, device=dev))
            t = timed(lambda: ops.gemm_split_tn(a, b))
            print(f"K={K:8d} dW[{Mo}, {No}]  {t:8.1f} us  {2.0 * K * Mo * No / t / 1e6:6.1f} TF  {(Mo + No) * K * 4 / t / 1e6:5.2f} TB/s")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "wgrad":
        wgrads()
    else:
        main()
