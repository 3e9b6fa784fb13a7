"""Diagnostic (libwf3d_stamp.so, -DWF3D_STAMP=1): where a wave's time goes inside gemm_split_x16_kernel."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wireframe-3d-prediction_amd"))
import torch  # noqa: E402
from wf3d import ops, _lib  # noqa: E402

lib = _lib.load()
lib.wf3d_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
dev = torch.device("cuda:0")
M = 131072
buf = (ctypes.c_ulonglong * 8)()
for K, N, tn in [(1024, 256, False), (512, 1024, False), (1024, 2048, False), (2048, 1024, False), (1024, 512, False), (2048, 1024, True)]:
    if tn:
        A, B = ops.split_rows(torch.randn(M, K, device=dev)), ops.split_rows(torch.randn(M, N, device=dev))
        fn = lambda: ops.gemm_split_tn(A, B)
    else:
        A, B = ops.split_rows(torch.randn(M, K, device=dev)), ops.split_rows(torch.randn(N, K, device=dev))
        out = torch.empty(M, N, device=dev)
        fn = lambda: ops.gemm_split(A, B, out=out)
    fn(); torch.cuda.synchronize()
    lib.wf3d_debug_stamps(buf, 1)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    lib.wf3d_debug_stamps(buf, 1)
    comp, dma, bar, n, epi, loop, waves, dma4 = [float(buf[i]) for i in range(8)]
    per = (comp + dma + bar) / n
    print(f"{'TN' if tn else 'NT'} K={K} N={N}: per slice {per:7.0f} ticks = compute {comp / n:7.0f} ({100 * comp / (comp + dma + bar):4.1f}%) "
          f"+ dma-wait {dma / n:6.0f} ({100 * dma / (comp + dma + bar):4.1f}%) + barrier {bar / n:6.0f} ({100 * bar / (comp + dma + bar):4.1f}%); "
          f"of which waiting for A(kt+1) after the first 4 slices of a tile {dma4 / max(n - 4 * waves, 1):6.0f}/slice; "
          f"per wave: loop {loop / waves:9.0f}  epilogue {epi / waves:8.0f} ticks ({100 * epi / (epi + loop):4.1f}% of loop+epilogue)")
