"""Row f-2, second half: the per-step bookkeeping of the reference's training loop without its
per-step host syncs.

train.py:145-157 calls `total_loss.item()` (twice) and copies sample 0's predicted and target vertices to the host on EVERY
step to keep a loss history, the best loss so far and a monitoring RMSE; each of these drains the GPU queue.  `TrainMeter`
keeps the same quantities in a small device-resident record, updated by one launch per step (csrc/loss.hip,
wf3d_meter_update), and `read()` brings everything back in one copy — call it when the loop logs (every 20 steps in the
reference), not every step.

    meter = TrainMeter(device)
    for step in range(n):
        out = model(x, counts); res = criterion(out, targets); res["total_loss"].backward(); opt.step()
        meter.update(res, out["vertices"], targets["vertices"], counts)          # no sync
        if step % 20 == 0:
            m = meter.read()                                                     # one device -> host copy
            log(m["total_loss"], m["vertex_rmse"], m["best_loss"], m["best_vertex_rmse"])
"""
import torch

from . import _lib
from ._lib import check
from .ops import _p, _stream


class TrainMeter:
    def __init__(self, device, history=4096):
        self.capacity = int(history)
        self.state = torch.zeros(8 + self.capacity, dtype=torch.float32, device=device)

    @torch.no_grad()
    def update(self, loss_dict, pred_vertices, target_vertices, counts=None):
        """loss_dict: WireframeLoss output (total_loss, vertex_loss, existence_loss, edge_loss: device scalars);
        pred_vertices [B, V, 3] (any vertex stride, unit inner stride), target_vertices [B, Vt, 3], counts [B] int64 or None."""
        def scalar(k):
            t = loss_dict.get(k)
            if t is None:
                return None
            if not t.is_cuda:
                raise RuntimeError("wf3d.TrainMeter: loss values must be device tensors (no CPU fallback)")
            return t.detach().reshape(-1).float()
        tot = scalar("total_loss")
        if tot is None:
            raise KeyError("total_loss")
        keep = [tot] + [scalar(k) for k in ("vertex_loss", "existence_loss", "edge_loss")]
        pv, tv = pred_vertices.detach(), target_vertices.detach()
        if pv.dtype != torch.float32 or tv.dtype != torch.float32 or pv.stride(-1) != 1 or tv.stride(-1) != 1:
            raise RuntimeError("wf3d.TrainMeter: fp32 vertices with unit inner stride expected")
        max_v = min(pv.shape[1], tv.shape[1])
        cnt = None
        if counts is not None:
            cnt = counts if counts.dtype == torch.int64 else counts.long()
            if not cnt.is_cuda:
                cnt = cnt.to(pv.device)
        check(_lib.load().wf3d_meter_update(_p(keep[0]), _p(keep[1]), _p(keep[2]), _p(keep[3]), _p(pv), pv.stride(1), _p(tv),
                                            tv.stride(1), _p(cnt), max_v, _p(self.state), self.capacity, _stream()), "meter_update")

    def read(self):
        """One device -> host copy: dict(steps, best_loss, best_vertex_rmse, total_loss, vertex_loss, existence_loss,
        edge_loss, vertex_rmse, loss_history (oldest first, at most `history` entries))."""
        s = self.state.cpu()
        steps = int(s[0])
        n = min(steps, self.capacity)
        ring = s[8:8 + self.capacity]
        if steps <= self.capacity:
            hist = ring[:n]
        else:
            k = steps % self.capacity
            hist = torch.cat([ring[k:], ring[:k]])
        return {"steps": steps, "best_loss": float(s[1]), "best_vertex_rmse": float(s[2]), "total_loss": float(s[3]),
                "vertex_loss": float(s[4]), "existence_loss": float(s[5]), "edge_loss": float(s[6]), "vertex_rmse": float(s[7]),
                "loss_history": hist.tolist()}
