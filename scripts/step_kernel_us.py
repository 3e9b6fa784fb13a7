#!/usr/bin/env python3
"""Average duration of the named kernels inside cfg2 steps, from HIP events around the ops (no profiler):
   python scripts/step_kernel_us.py pool4_fwd pool4_bwd ...   (names of wf3d.ops functions)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wireframe-3d-prediction_amd"))
from models.PointCloudToWireframe import PointCloudToWireframe  # noqa: E402
from wf3d import ops  # noqa: E402
import wf3d.functional as F  # noqa: E402

dev = torch.device("cuda:0")
names = sys.argv[1:] or ["pool4_fwd", "pool4_bwd"]
rec = {n: [] for n in names}


def wrap(n):
    f = getattr(ops, n)

    def g(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = f(*a, **k)
        e1.record()
        rec[n].append((e0, e1))
        return r
    setattr(ops, n, g)
    if hasattr(F.ops, n):
        setattr(F.ops, n, g)


for n in names:
    wrap(n)
torch.manual_seed(1)
m = PointCloudToWireframe(input_dim=8, max_vertices=64).to(dev)
m.vertex_predictor.ensure_point_pool_proj(1024, dev)
m.set_dropout(0.1)
m.train()
x = torch.randn(32, 4096, 8, device=dev)
counts = torch.full((32,), 64, dtype=torch.long, device=dev)
for it in range(14):
    if it == 4:
        torch.cuda.synchronize()
        for n in names:
            rec[n].clear()
    m.zero_grad(set_to_none=True)
    out = m(x, counts)
    (out["vertices"].sum() + out["existence_probabilities"].sum() + out["edge_probs"].sum()).backward()
torch.cuda.synchronize()
for n in names:
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in rec[n])
    print(f"{n:24s} {len(ts):4d} calls  median {ts[len(ts) // 2]:8.1f} us  min {ts[0]:8.1f} us")
