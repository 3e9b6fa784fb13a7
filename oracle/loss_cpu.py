"""CPU oracle for row f-1: the reference's WireframeLoss
(losses/WireframeLoss.py:38-283) restated functionally.  TEST INFRASTRUCTURE ONLY
(same rules as oracle/reference_cpu.py).  Pinned by tests/golden/loss_*.npz, produced by
tests/golden/make_golden_loss.py from the imported reference class.

cost[p, t] = sum_k |v_p - t_t|_k + |e_p - 1|   for real target columns t < count
           = e_p                                for the V - count dummy columns      (:142-219)
Hungarian assignment (scipy.optimize.linear_sum_assignment, as the reference, :236), matches
with a real target kept; vertex loss = SmoothL1 over the matched pairs averaged over all
matches of the batch (:260-283); existence / edge losses = nn.BCELoss means (:72-88).
"""
import torch
import torch.nn.functional as F
from scipy.optimize import linear_sum_assignment


def cost_matrix(pred_v, pred_e, tgt_v, count):
    V = pred_v.shape[0]
    c = torch.cdist(pred_v, tgt_v[:count], p=1) + (pred_e.unsqueeze(1) - 1.0).abs()
    if V - count > 0:
        c = torch.cat([c, pred_e.unsqueeze(1).expand(-1, V - count)], dim=1)
    return c


def hungarian(predictions, targets):
    out = []
    pv, pe = predictions["vertices"], predictions["existence_probabilities"]
    for b in range(pv.shape[0]):
        count = int(targets["vertex_counts"][b].item())
        c = cost_matrix(pv[b], pe[b], targets["vertices"][b], count).detach().cpu().numpy()
        pi, ti = linear_sum_assignment(c)
        keep = ti < count
        out.append((pi[keep], ti[keep]))
    return out


def wireframe_loss(predictions, targets, vertex_weight=1.0, edge_weight=1.0, existence_weight=1.0):
    pv = predictions["vertices"]
    matches = hungarian(predictions, targets)
    tot, n = 0.0, 0
    for b, (pi, ti) in enumerate(matches):
        if len(pi):
            tot = tot + F.smooth_l1_loss(pv[b, pi], targets["vertices"][b, ti]) * len(pi)
            n += len(pi)
    vertex_loss = tot / n if n else torch.tensor(0.0)
    existence_loss = F.binary_cross_entropy(predictions["existence_probabilities"], targets["vertex_existence"].float())
    pe, tl = predictions["edge_probs"], targets["edge_labels"]
    m = min(pe.shape[1], tl.shape[1]) if pe.numel() and tl.numel() else 0
    edge_loss = F.binary_cross_entropy(pe[:, :m], tl[:, :m]) if m > 0 else torch.tensor(0.0)
    total = vertex_weight * vertex_loss + existence_weight * existence_loss + edge_weight * edge_loss
    return {"total_loss": total, "vertex_loss": vertex_loss, "existence_loss": existence_loss,
            "edge_loss": edge_loss}, matches
