#!/usr/bin/env python3
"""4-way pool forward / backward at cfg2 and cfg4 shapes: time and achieved HBM GB/s (csrc/pool.hip)."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wireframe-3d-prediction_amd"))
import torch
from wf3d import ops
dev = torch.device("cuda:0")
flush = torch.empty(1 << 29, dtype=torch.uint8, device=dev)
for B, N, C in [(32, 4096, 512), (8, 16384, 512)]:
    pf = torch.randn(B, N, C, device=dev)
    valid = (torch.rand(B * N, device=dev) > 0.1).float()
    ts = []
    for _ in range(12):
        flush.zero_()                                   # evict the Infinity Cache: the pool reads a tensor the GEMM just streamed out
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); po = ops.pool4_fwd(pf, valid, packed=True); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    t = statistics.median(ts[2:])
    print(f"pool4_fwd B={B} N={N} C={C}: {t * 1e3:.1f} us  {pf.numel() * 4 / t / 1e6:.0f} GB/s (partial + final kernels)")
