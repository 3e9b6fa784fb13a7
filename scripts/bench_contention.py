#!/usr/bin/env python3
"""What a few busy CUs cost the split GEMMs.  A stand-in for a collective's kernels (scripts/micro/cu_hog.hip: `k`
workgroups that hold a CU each for a few milliseconds) runs on a second stream while the four encoder forward GEMMs and
their wgrads are timed on the first.  Build the hog first:
  hipcc -O3 --offload-arch=gfx950 -shared -fPIC scripts/micro/cu_hog.hip -o scripts/micro/build/libcu_hog.so"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wireframe-3d-prediction_amd"))
from wf3d import ops  # noqa: E402

hog = ctypes.CDLL(os.path.join(ROOT, "scripts", "micro", "build", "libcu_hog.so"))
hog.cu_hog.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda:0")
M = 131072
side = torch.cuda.Stream(dev)


def timed(fn, k, reps=5):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        if k:
            hog.cu_hog(k, 6000, ctypes.c_void_p(side.cuda_stream))
            torch.cuda._sleep(200000)                    # let the hog's workgroups take their CUs first
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]


def main():
    print(f"{'GEMM':28s}" + "".join(f"  busy CUs={k:3d}" for k in (0, 1, 8, 32)))
    for K, N in ((512, 1024), (1024, 2048), (2048, 1024), (1024, 512)):
        X, W, G = ops.split_rows(torch.randn(M, K, device=dev)), ops.split_rows(torch.randn(N, K, device=dev)), ops.split_rows(torch.randn(M, N, device=dev))
        out, dw = torch.empty(M, N, device=dev), torch.empty(N, K, device=dev)
        for name, fn in ((f"forward K={K} N={N}", lambda: ops.gemm_split(X, W, out=out)),
                         (f"wgrad   {N}x{K}", lambda: ops.gemm_split_tn(G, X, out=dw))):
            fn()
            print(f"{name:28s}" + "".join(f"  {timed(fn, k):10.1f} us" for k in (0, 1, 8, 32)))


if __name__ == "__main__":
    main()
