#!/usr/bin/env python3
"""Golden fixture for row f-1: run the REFERENCE's losses.WireframeLoss (build container only) on
detgen-generated predictions/targets and store its four loss values, its Hungarian matches and the
gradients it sends back into the model outputs.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_loss.py
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
from losses.WireframeLoss import WireframeLoss          # the reference


def case(tag, B, V, counts, max_e_pred, max_e_tgt, seed, weights=(3.0, 1.5, 1.0)):
    """weights = (vertex, edge, existence) as train.py:90-94 passes them (3.0 / 1.5 / 1.0)."""
    pv = torch.from_numpy(detgen.normalish(tag + ".pv", (B, V, 3), seed)).requires_grad_()
    pe = torch.sigmoid(torch.from_numpy(2.0 * detgen.normalish(tag + ".pe", (B, V), seed))).requires_grad_()
    pp = torch.sigmoid(torch.from_numpy(2.0 * detgen.normalish(tag + ".pp", (B, max_e_pred), seed))).requires_grad_()
    tv = torch.from_numpy(detgen.normalish(tag + ".tv", (B, V, 3), seed))
    cnt = torch.tensor(counts, dtype=torch.long)
    te = (torch.arange(V)[None, :] < cnt[:, None]).float()
    tl = (torch.from_numpy(detgen.uniform(tag + ".tl", (B, max_e_tgt), 0, 1, seed)) > 0.7).float()
    crit = WireframeLoss(vertex_weight=weights[0], edge_weight=weights[1], existence_weight=weights[2])
    preds = {"vertices": pv, "existence_probabilities": pe, "edge_probs": pp}
    tgts = {"vertices": tv, "vertex_existence": te, "edge_labels": tl, "vertex_counts": cnt}
    out = crit(preds, tgts)
    out["total_loss"].backward()
    matches = crit._hungarian_matching(preds, tgts)
    res = {"meta.B": np.array(B), "meta.V": np.array(V), "meta.counts": np.array(counts), "meta.seed": np.array(seed),
           "meta.max_e_pred": np.array(max_e_pred), "meta.max_e_tgt": np.array(max_e_tgt),
           "meta.weights": np.array(weights)}
    for k in ("total_loss", "vertex_loss", "existence_loss", "edge_loss"):
        res["out." + k] = np.array(float(out[k]))
    res["grad.vertices"] = pv.grad.numpy()
    res["grad.existence"] = pe.grad.numpy()
    res["grad.edge_probs"] = pp.grad.numpy()
    res["match.lens"] = np.array([len(m[0]) for m in matches])
    res["match.pred"] = np.concatenate([np.asarray(m[0]) for m in matches]).astype(np.int64)
    res["match.tgt"] = np.concatenate([np.asarray(m[1]) for m in matches]).astype(np.int64)
    np.savez_compressed(os.path.join(HERE, tag + ".npz"), **res)
    print(tag, {k: float(out[k]) for k in out})


if __name__ == "__main__":
    case("loss_a", 3, 26, [26, 4, 24], 325, 325, 21)            # the real first batch's counts
    case("loss_b", 4, 16, [16, 16, 1, 9], 100, 120, 22)         # edge widths differ: truncation to the common size
    case("loss_c", 2, 8, [0, 8], 28, 28, 23, weights=(1.0, 1.0, 1.0))   # a sample with no target vertex
