"""Row f-4: evaluation post-processing behind the hot path (reference evaluate.py:74-110) on the device.

    samples = wireframes_from_predictions(model(x))        # ONE kernel + ONE device->host copy for the whole batch
    evaluate_batch(predictions, wf_vertices, wf_edges, ap_calculator)      # the reference's per-sample loop, fed from it

`hausdorff_lines` is the device form of eval.ap_calculator.hausdorff_distance_line (csrc/evalpost.hip)."""
import numpy as np
import torch

from . import _lib
from ._lib import check
from .functional import edge_index_lists
from .ops import _stream


def hausdorff_lines(p_line, t_line, sample_points, device):
    """[N, M] float64 numpy matrix; p_line [N, 2, 3], t_line [M, 2, 3] numpy arrays (any float dtype).  The segment
    directions are formed in the arrays' common dtype, as the reference's concatenate + subtraction does (:22-26)."""
    N, M = p_line.shape[0], t_line.shape[0]
    both = np.concatenate((p_line, t_line), axis=0)
    start = both[:, 0, :].astype(np.float64)
    diff = (both[:, 1, :] - both[:, 0, :]).astype(np.float64)
    buf = torch.from_numpy(np.ascontiguousarray(np.stack((start, diff)))).to(device)          # [2, N + M, 3]
    out = torch.empty(N, M, dtype=torch.float64, device=device)
    ps, pd = buf[0, :N], buf[1, :N]
    ts, td = buf[0, N:], buf[1, N:]
    check(_lib.load().wf3d_hausdorff_lines(ps.data_ptr(), pd.data_ptr(), ts.data_ptr(), td.data_ptr(), N, M, int(sample_points),
                                           out.data_ptr(), _stream()), "hausdorff_lines")
    return out.cpu().numpy()


def wireframes_from_predictions(predictions, threshold=0.5):
    """Per sample: dict(pred_vertices [V, 3] float32, pd_edges [k, 2] int64, pd_edges_vertices [k, 2, 3] float32) — the
    edges whose probability exceeds `threshold`, end points ordered higher-z first (evaluate.py:74-92)."""
    verts, probs = predictions["vertices"], predictions["edge_probs"]
    B, V, _ = verts.shape
    lists = predictions["edge_indices"]
    counts = [int(round((1 + (1 + 8 * len(e)) ** 0.5) / 2)) if len(e) else 0 for e in lists]
    max_e = probs.shape[1]
    dev = verts.device
    v = verts.detach().float()
    if v.stride(2) != 1:
        v = v.contiguous()
    ev = torch.zeros(B, max_e, 2, 3, dtype=torch.float32, device=dev)
    keep = torch.zeros(B, max_e, dtype=torch.uint8, device=dev)
    cnt = torch.tensor(counts, dtype=torch.int32).to(dev)
    pr = probs.detach().float().contiguous()
    check(_lib.load().wf3d_edge_endpoints(v.data_ptr(), v.stride(0), v.stride(1), cnt.data_ptr(), pr.data_ptr(), B, V, max_e,
                                          float(threshold), ev.data_ptr(), keep.data_ptr(), _stream()), "edge_endpoints")
    # one packed device->host transfer for the batch
    packed = torch.cat([v.reshape(B, -1) if v.is_contiguous() else v.contiguous().reshape(B, -1), ev.reshape(B, -1), keep.float()], dim=1).cpu().numpy()
    nv, ne = V * 3, max_e * 6
    out = []
    for s in range(B):
        row = packed[s]
        mask = row[nv + ne:] > 0.5
        idx = np.array(lists[s], dtype=np.int64).reshape(-1, 2)
        m = mask[:len(idx)]
        out.append({"pred_vertices": row[:nv].reshape(V, 3).copy(),
                    "pd_edges": idx[m],
                    "pd_edges_vertices": row[nv:nv + ne].reshape(max_e, 2, 3)[:len(idx)][m].copy()})
    return out


def _gt_edge_vertices(gt_vertices, gt_edges):
    if len(gt_edges) == 0:
        return np.empty((0, 2, 3))
    ev = np.stack((gt_vertices[gt_edges[:, 0]], gt_vertices[gt_edges[:, 1]]), axis=1)
    return ev[np.arange(ev.shape[0])[:, np.newaxis], np.flip(np.argsort(ev[:, :, -1]), axis=1)]


def evaluate_batch(predictions, wf_vertices, wf_edges, ap_calculator, threshold=0.5):
    """The body of the reference's evaluation loop (evaluate.py:74-110) for one batch: `wf_vertices` / `wf_edges` are the
    collate_batch lists of tensors; accumulates into `ap_calculator`."""
    for s, w in enumerate(wireframes_from_predictions(predictions, threshold)):
        gt_v = wf_vertices[s].numpy() if torch.is_tensor(wf_vertices[s]) else np.asarray(wf_vertices[s])
        gt_e = (wf_edges[s].numpy() if torch.is_tensor(wf_edges[s]) else np.asarray(wf_edges[s])).astype(np.int64)
        pd_ev = w["pd_edges_vertices"] if len(w["pd_edges"]) > 0 else np.empty((0, 2, 3))
        batch = {"predicted_vertices": w["pred_vertices"][np.newaxis, :],
                 "predicted_edges": w["pd_edges"][np.newaxis, :],
                 "pred_edges_vertices": pd_ev.reshape((1, -1, 2, 3)),
                 "wf_vertices": gt_v[np.newaxis, :],
                 "wf_edges": gt_e[np.newaxis, :],
                 "wf_edges_vertices": _gt_edge_vertices(gt_v, gt_e).reshape((1, -1, 2, 3))}
        ap_calculator.compute_metrics(batch)
