#!/usr/bin/env python3
"""Fixture generator for row f-4 (evaluation post-processing): imports the REFERENCE's eval/ap_calculator.py from
/root/reference (build container only; numpy + scipy) and records, for synthetic predictions built around a real demo
wireframe, what its hausdorff_distance_line and APCalculator return.  The per-sample batch assembly of evaluate.py:74-110
(a function that also loads datasets and a checkpoint, so it cannot be called) is restated here in a few numpy lines.
Output: tests/golden/eval.npz (numeric only).   python tests/golden/make_golden_eval.py"""
import contextlib
import importlib.util
import io
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("ref_ap", "/root/reference/eval/ap_calculator.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)


def load_obj(path):
    v, e = [], []
    for line in open(path):
        t = line.split()
        if t and t[0] == "v":
            v.append([float(x) for x in t[1:4]])
        elif t and t[0] == "l":
            a, b = int(t[1]) - 1, int(t[2]) - 1
            e.append((min(a, b), max(a, b)))
    return np.array(v), np.array(sorted(set(e)), dtype=np.int64)


def z_first(vertices, edges):
    if len(edges) == 0:
        return np.empty((0, 2, 3))
    ev = np.stack((vertices[edges[:, 0]], vertices[edges[:, 1]]), axis=1)
    return ev[np.arange(len(ev))[:, None], np.flip(np.argsort(ev[:, :, -1]), axis=1)]


def sample_batch(pred_vertices, edge_indices, edge_probs, gt_v, gt_e):
    """One sample as evaluate.py assembles it (float32 predictions, float32 labels)."""
    mask = edge_probs > 0.5
    pd_edges = np.array(edge_indices)[mask]
    return {"predicted_vertices": pred_vertices[None], "predicted_edges": pd_edges[None],
            "pred_edges_vertices": z_first(pred_vertices, pd_edges).reshape(1, -1, 2, 3),
            "wf_vertices": gt_v[None], "wf_edges": gt_e[None], "wf_edges_vertices": z_first(gt_v, gt_e).reshape(1, -1, 2, 3)}


rng = np.random.RandomState(2024)
gt_v64, gt_e = load_obj(os.path.join(HERE, "building3d", "train", "wireframe", "100.obj"))
gt_v = (gt_v64 - gt_v64.mean(0)).astype(np.float32)
n = len(gt_v)
V = n + 4
E = V * (V - 1) // 2
pairs = np.array([[i, j] for i in range(V) for j in range(i + 1, V)], dtype=np.int64)
out = {"gt_v": gt_v, "gt_e": gt_e, "V": np.array(V)}
cases = []
# case 0: good prediction — label vertices + small noise, label edges likely, a few spurious ones
pv = np.concatenate([gt_v + rng.normal(0, 0.08, gt_v.shape), rng.uniform(-8, 8, (V - n, 3))]).astype(np.float32)
is_gt = np.array([any((g == p).all() for g in gt_e) for p in pairs])
pr = np.where(is_gt, rng.uniform(0.55, 0.99, E), rng.uniform(0.0, 0.45, E)).astype(np.float32)
flip = rng.choice(E, 6, replace=False)
pr[flip] = 1.0 - pr[flip]
cases.append((pv, pr))
# case 1: noisier vertices, fewer edges
pv = np.concatenate([gt_v + rng.normal(0, 0.6, gt_v.shape), rng.uniform(-8, 8, (V - n, 3))]).astype(np.float32)
pr = np.where(is_gt, rng.uniform(0.3, 0.9, E), rng.uniform(0.0, 0.2, E)).astype(np.float32)
cases.append((pv, pr))
# case 2: no edge above threshold -> the corners-only branch
cases.append((pv.copy(), (pr * 0.4).astype(np.float32)))
calc = ref.APCalculator(distance_thresh=1)
for c, (pv, pr) in enumerate(cases):
    out[f"c{c}.vertices"], out[f"c{c}.probs"] = pv, pr
    b = sample_batch(pv.copy(), pairs, pr, gt_v.copy(), gt_e.copy())
    if c == 0:
        out["hd.p"], out["hd.t"] = b["pred_edges_vertices"][0].copy(), b["wf_edges_vertices"][0].copy()
        out["hd.matrix"] = ref.hausdorff_distance_line(b["pred_edges_vertices"][0].copy(), b["wf_edges_vertices"][0].copy())
    with contextlib.redirect_stdout(io.StringIO()):
        calc.compute_metrics(b)
    out[f"c{c}.counters"] = np.array([float(calc.ap_dict[k]) for k in ("tp_corners", "tp_fp_corners", "tp_fn_corners", "distance",
                                                                         "tp_edges", "wed", "tp_fp_edges", "tp_fn_edges")])
with contextlib.redirect_stdout(io.StringIO()):
    calc.output_accuracy()
keys = ["average_corner_offset", "average_wed", "corners_precision", "corners_recall", "corners_f1", "edges_precision", "edges_recall", "edges_f1"]
out["final"] = np.array([float(calc.ap_dict[k]) for k in keys])
out["pairs"] = pairs
# pure-function fixtures
a = rng.normal(0, 1, (7, 3)); b = np.concatenate([a[[1, 4]], rng.normal(0, 1, (3, 3))])
out["rc.a"], out["rc.b"], out["rc.out"] = a, b, ref.remove_corners(a, b)
ev = rng.normal(0, 1, (5, 2, 3)); verts = np.concatenate([ev.reshape(-1, 3)[::2], rng.normal(0, 1, (2, 3))])
out["ce.edges"], out["ce.verts"], out["ce.out"] = ev, verts, ref.computer_edges(ev, verts)
np.savez_compressed(os.path.join(HERE, "eval.npz"), **out)
print("wrote eval.npz:", {k: out[k].tolist() for k in out if k.endswith("counters")}, out["final"].tolist())
