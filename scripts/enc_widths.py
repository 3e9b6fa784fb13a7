#!/usr/bin/env python3
"""PointNetEncoder at widths other than the model's: bf16x3 against fp32 mode, per parameter gradient (max |diff| / max |ref|).
   python scripts/enc_widths.py 264 520 264     (hidden dims..., output dim)"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "wireframe-3d-prediction_amd"))
from models.PointNetEncoder import PointNetEncoder  # noqa: E402
from wf3d import config  # noqa: E402

dev = torch.device("cuda:0")
dims = [int(a) for a in sys.argv[1:]] or [264, 520, 264]
hidden, out = dims[:-1], dims[-1]
res = {}
for prec in ("fp32", "bf16x3"):
    config.set_precision(prec)
    torch.manual_seed(5)
    enc = PointNetEncoder(8, hidden, out).to(dev)
    g0 = torch.Generator().manual_seed(9)
    x = torch.randn(3, 1024, 8, generator=g0).to(dev)
    cg, cp = torch.randn(3, out, generator=g0).to(dev), (torch.randn(3, 1024, out, generator=g0) * 0.01).to(dev)
    g, pf = enc(x)
    ((g * cg).sum() + (pf * cp).sum()).backward()
    res[prec] = {"g": g.detach(), "pf": pf.detach(), **{n: p.grad.detach() for n, p in enc.named_parameters()}}
for k in res["fp32"]:
    a, b = res["bf16x3"][k].double(), res["fp32"][k].double()
    print(f"{k:28s} {tuple(a.shape)!s:16s} {float((a - b).abs().max() / b.abs().max().clamp_min(1e-30)):.2e}")
