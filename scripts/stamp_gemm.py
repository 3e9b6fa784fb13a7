"""Slice-by-slice timeline of the persistent split GEMM (diagnostic build -DWF3D_STAMP=1, scripts/build_variants.sh):
every workgroup records s_memtime at the start of each k32 slice.  Prints, per shape, the median steady-state slice
time and the extra time of the slices around a tile switch.

    scripts/build_variants.sh gemm_split.hip stamp "-DWF3D_STAMP=1"
    WF3D_LIB=wireframe-3d-prediction_amd/libwf3d_stamp.so python scripts/stamp_gemm.py
"""
import ctypes
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wireframe-3d-prediction_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from wf3d import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
M = 131072
PER = 1024
lib = _lib.load()
lib.wf3d_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
for K, N in [(512, 1024), (1024, 2048), (2048, 1024), (1024, 512)]:
    X = torch.randn(M, K, device=dev)
    W = torch.randn(N, K, device=dev) * 0.05
    b = torch.randn(N, device=dev)
    Xs, Ws = ops.split_rows(X), ops.split_rows(W)
    out = torch.empty(M, N, device=dev)
    for bias in (None, b):
        for _ in range(3):
            ops.gemm_split(Xs, Ws, bias=bias, out=out)
        torch.cuda.synchronize()
        buf = np.zeros(256 * PER, dtype=np.uint64)
        assert lib.wf3d_debug_stamps(buf.ctypes.data, buf.size) == 0
        st = buf.reshape(256, PER).astype(np.int64)
        kt = K // 32
        ntile = (M // 256) * (N // 256) // 256
        ns = min(kt * ntile, PER)
        k0 = ns - (st[:, :ns] > 0).sum(axis=1)                  # dephase experiments: a workgroup's stream is shorter by its offset
        ns_wg = int((st[:, :ns] > 0).sum(axis=1).min())
        d = np.diff(st[:, :ns_wg], axis=1)                       # [wg, slice] duration in shader cycles
        med = float(np.median(d))
        # the slices that carry a tile's stores are the (tiles - 1) longest of every workgroup
        top = np.sort(d, axis=1)[:, -(ntile - 1):]
        extra = top - med
        rest = np.sort(d, axis=1)[:, :-(ntile - 1)]
        wall = (st[np.arange(256), ns_wg - 1] - st[:, 0])
        print(f"K={K} N={N} bias={'y' if bias is not None else 'n'}: tiles/CU {ntile}, slices/tile {kt}; slice median {med:.0f} cycles "
              f"(MFMA-only 3072); store slices: mean extra {extra.mean():.0f} cycles (p10 {np.percentile(extra, 10):.0f}, p90 {np.percentile(extra, 90):.0f}); "
              f"other slices mean {rest.mean():.0f}; stream mean {wall.mean():.0f} cycles, of which store-slice extra {100 * extra.sum(axis=1).mean() / wall.mean():.1f} %, "
              f"other excess over 3072 {100 * (rest.sum(axis=1).mean() - 3072 * rest.shape[1]) / wall.mean():.1f} %")
