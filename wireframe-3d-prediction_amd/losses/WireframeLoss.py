"""Drop-in `losses.WireframeLoss.WireframeLoss` (reference losses/WireframeLoss.py:6-283) — row f-1,
the caller of the hot path's backward.

Same constructor, same `forward(predictions, targets) -> dict` with the same four keys.  The
Hungarian assignment is still scipy's `linear_sum_assignment` on the host (as in the reference,
:236), but all B cost matrices come from ONE kernel and ONE device->host copy (the reference does a
`.cpu().numpy()` sync per sample), and SmoothL1 / BCE / BCE plus their gradients w.r.t. the model
outputs are one more kernel; `total_loss.backward()` hands those gradients to the model."""
import numpy as np
import torch
import torch.nn as nn
from scipy.optimize import linear_sum_assignment

from wf3d import ops


class _LossDevFn(torch.autograd.Function):
    """Fully asynchronous variant: cost matrices, assignment, loss terms and gradients all on the device."""

    @staticmethod
    def forward(ctx, verts, exist, edge, tverts, texist, tlabel, counts, weights):
        exist_c = exist.contiguous()
        cost = ops.loss_cost_matrix(verts, exist_c, tverts, counts)
        # the square problem, rows in prediction order like scipy: L1 costs have exact ties (2 of 32 random samples
        # at V = 64) and the rectangular shortcut of ops.loss_assign(cost, counts) resolves them differently
        col4row = ops.loss_assign(cost)
        losses, dv, de, dd = ops.loss_terms_assigned(verts, exist_c, edge.contiguous(), tverts, texist, tlabel,
                                                     col4row, counts, weights)
        ctx.save_for_backward(dv, de, dd)
        ctx.mark_non_differentiable(losses)
        return losses[3].clone(), losses

    @staticmethod
    def backward(ctx, g_total, _g_losses):
        dv, de, dd = ctx.saved_tensors
        return (dv * g_total, de * g_total, dd * g_total, None, None, None, None, None)


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, verts, exist, edge, tverts, texist, tlabel, m_pred, m_tgt, m_off, n_match, weights):
        losses, dv, de, dd = ops.loss_terms(verts, exist.contiguous(), edge.contiguous(), tverts, texist, tlabel,
                                            m_pred, m_tgt, m_off, n_match, weights)
        ctx.save_for_backward(dv, de, dd)
        ctx.mark_non_differentiable(losses)
        return losses[3].clone(), losses

    @staticmethod
    def backward(ctx, g_total, _g_losses):
        dv, de, dd = ctx.saved_tensors
        return (dv * g_total, de * g_total, dd * g_total, None, None, None, None, None, None, None, None)


class WireframeLoss(nn.Module):
    def __init__(self, vertex_weight=1.0, edge_weight=1.0, existence_weight=1.0, assignment="device"):
        """assignment: "device" (default) = Hungarian matching on the GPU, no host sync at all;
        "scipy" = the reference's scipy.optimize.linear_sum_assignment on the host (one sync)."""
        super().__init__()
        if assignment not in ("device", "scipy"):
            raise ValueError("assignment must be 'device' or 'scipy'")
        self.assignment = assignment
        self.vertex_weight = vertex_weight
        self.edge_weight = edge_weight
        self.existence_weight = existence_weight
        self.smooth_l1_loss = nn.SmoothL1Loss()      # kept for attribute parity; not called
        self.bce_loss = nn.BCELoss()

    def _hungarian_matching(self, predictions, targets):
        """list of (pred_indices, target_indices) numpy arrays per sample (reference :106-237)."""
        pv = predictions["vertices"]
        B, V, _ = pv.shape
        counts = targets["vertex_counts"].to(device=pv.device, dtype=torch.int64).contiguous()
        tv = targets["vertices"].to(device=pv.device, dtype=torch.float32).contiguous()
        cost = ops.loss_cost_matrix(pv.detach(), predictions["existence_probabilities"].detach().contiguous(), tv, counts)
        cost_np = cost.cpu().numpy()                               # the one sync of the loss
        cnt = counts.cpu().tolist()
        out = []
        for b in range(B):
            if cnt[b] > V:
                raise ValueError("target vertex count exceeds max_vertices (the reference's inf-padded matrix is infeasible too)")
            pi, ti = linear_sum_assignment(cost_np[b])
            keep = ti < cnt[b]
            out.append((pi[keep], ti[keep]))
        return out

    def forward(self, predictions, targets):
        pv = predictions["vertices"]
        pe = predictions["existence_probabilities"]
        pp = predictions["edge_probs"]
        dev = pv.device
        if self.assignment == "device":
            counts = targets["vertex_counts"].to(device=dev, dtype=torch.int64).contiguous()
            tv = targets["vertices"].to(device=dev, dtype=torch.float32).contiguous()
            te = targets["vertex_existence"].to(device=dev, dtype=torch.float32).contiguous()
            tl = targets["edge_labels"].to(device=dev, dtype=torch.float32).contiguous()
            w = (self.vertex_weight, self.existence_weight, self.edge_weight)
            total, parts = _LossDevFn.apply(pv, pe, pp, tv, te, tl, counts, w)
            return {"total_loss": total, "vertex_loss": parts[0], "existence_loss": parts[1], "edge_loss": parts[2]}
        matches = self._hungarian_matching(predictions, targets)
        lens = [len(m[0]) for m in matches]
        off = np.zeros(len(lens) + 1, dtype=np.int32)
        np.cumsum(lens, out=off[1:])
        n_match = int(off[-1])
        cat = lambda k: (np.concatenate([m[k] for m in matches]).astype(np.int32) if n_match else np.zeros(0, np.int32))   # noqa: E731
        packed = torch.from_numpy(np.concatenate([cat(0), cat(1), off])).to(dev, non_blocking=True)
        m_pred, m_tgt, m_off = packed[:n_match], packed[n_match:2 * n_match], packed[2 * n_match:]
        tv = targets["vertices"].to(device=dev, dtype=torch.float32).contiguous()
        te = targets["vertex_existence"].to(device=dev, dtype=torch.float32).contiguous()
        tl = targets["edge_labels"].to(device=dev, dtype=torch.float32).contiguous()
        # weights in the kernel's (vertex, existence, edge) order
        w = (self.vertex_weight, self.existence_weight, self.edge_weight)
        total, parts = _LossFn.apply(pv, pe, pp, tv, te, tl, m_pred, m_tgt, m_off, n_match, w)
        return {"total_loss": total, "vertex_loss": parts[0], "existence_loss": parts[1], "edge_loss": parts[2]}
