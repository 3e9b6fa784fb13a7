#!/usr/bin/env python3
"""Golden fixture (SURVEY.md §8c case vi): three optimisation steps of the REFERENCE model under the REFERENCE
loss, in the order train.py:96-142 runs them (Adam created before the first forward, zero_grad / forward /
WireframeLoss / backward / clip_grad_norm_(1.0) / step), dropout zeroed.  Stored: the four loss values and the
pre-clip gradient norm of every step, and checksums of a few parameters after the last step.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_traj.py
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.environ.get("WF3D_REFERENCE", "/root/reference"))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from oracle import detgen
from losses.WireframeLoss import WireframeLoss                         # the reference
from models.PointCloudToWireframe import PointCloudToWireframe        # the reference

SEED, B, N, V, STEPS = 7, 3, 640, 12, 3
COUNTS = [12, 4, 9]


def inputs(seed=SEED):
    x = detgen.normalish("traj.x", (B, N, 8), seed)
    x[1, ::5] = 0.0                                                    # zero-padded points in one cloud
    cnt = np.array(COUNTS, dtype=np.int64)
    tv = 0.5 * detgen.normalish("traj.tv", (B, V, 3), seed)
    te = (np.arange(V)[None, :] < cnt[:, None]).astype(np.float32)
    tl = (detgen.uniform("traj.tl", (B, V * (V - 1) // 2), 0, 1, seed) > 0.75).astype(np.float32)
    return x, cnt, tv, te, tl


def zero_dropout(model):
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if isinstance(m, torch.nn.MultiheadAttention):
            m.dropout = 0.0


if __name__ == "__main__":
    torch.manual_seed(SEED)
    model = PointCloudToWireframe(input_dim=8, max_vertices=V)
    zero_dropout(model)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-6)            # train.py:96
    crit = WireframeLoss(vertex_weight=3.0, edge_weight=1.5, existence_weight=1.0)    # train.py:90-94
    x, cnt, tv, te, tl = (torch.from_numpy(a) for a in inputs())
    tgts = {"vertices": tv, "vertex_existence": te, "edge_labels": tl, "vertex_counts": cnt}
    model.train()
    rows = []
    for step in range(STEPS):
        opt.zero_grad()
        out = model(x, cnt)
        losses = crit(out, tgts)
        losses["total_loss"].backward()
        if step == 0:      # per-parameter view of the first step: gradient norms before clipping ...
            names = [k for k, _ in model.named_parameters()]
            g0 = np.array([float(p.grad.double().norm()) if p.grad is not None else -1.0 for _, p in model.named_parameters()])
        gn = torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.0)         # train.py:141
        before = [p.detach().clone() for p in model.parameters()] if step == 0 else None
        opt.step()
        if step == 0:      # ... and how far Adam moved each tensor (L1 of the update)
            d0 = np.array([float((p.detach() - b).double().abs().sum()) for p, b in zip(model.parameters(), before)])
            up0 = np.array([int(((p.detach() - b) > 0).sum()) for p, b in zip(model.parameters(), before)])     # elements moved up
        rows.append([float(losses[k]) for k in ("total_loss", "vertex_loss", "existence_loss", "edge_loss")] + [float(gn)])
        print(step, rows[-1])
    sd = model.state_dict()
    res = {"meta.seed": np.array(SEED), "meta.dims": np.array([B, N, V]), "meta.counts": np.array(COUNTS),
           "traj": np.array(rows, dtype=np.float64), "step0.names": np.array(names), "step0.grad_norm": g0,
           "step0.update_l1": d0, "step0.moved_up": up0}
    for k in ("encoder.mlp.0.weight", "encoder.mlp.16.bias", "vertex_predictor.final_layer.weight",
              "edge_predictor.edge_mlp.10.weight", "vertex_predictor.point_pool_proj.weight"):
        t = sd[k].double()
        res["after." + k] = np.array([float(t.sum()), float(t.abs().sum()), float((t * t).sum())])
    np.savez_compressed(os.path.join(HERE, "traj.npz"), **res)
