"""Row f-2: the tail of the reference's training step (train.py:141-142),

    torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.0)
    optimizer.step()                                   # Adam(lr=1e-3, weight_decay=1e-6), train.py:96

as two kernel launches over all parameter tensors (csrc/optim.hip) instead of ~500 small foreach kernels.

    opt = ClipAdam(model.parameters(), lr=1e-3, weight_decay=1e-6, max_norm=1.0, norm_params=model.parameters)
    ...
    loss.backward()
    opt.step()              # clip (in place, over norm_params) + Adam (over the optimizer's own parameters)

`norm_params` is a callable so that parameters created AFTER the optimizer — the reference's lazy
`vertex_predictor.point_pool_proj`, SURVEY.md §9 Q1 — enter the global norm and get their gradient scaled exactly as
`clip_grad_norm_(model.parameters())` does, while never being updated (they are not in the optimizer, as in train.py).
State (`step`, `exp_avg`, `exp_avg_sq`) uses torch.optim.Adam's names, so `state_dict()` round-trips with it."""
import ctypes

import torch

from . import _lib
from ._lib import check
from .ops import _stream


class ClipAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_norm=None, norm_params=None):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError("ClipAdam: bad hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        if len(self.param_groups) != 1:
            raise ValueError("ClipAdam: one parameter group (the global gradient norm spans all tensors of one launch)")
        self.max_norm = max_norm
        self._norm_params = norm_params
        self.last_grad_norm = None          # device scalar: pre-clip global L2 norm of the last step

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        group = self.param_groups[0]
        own = [p for p in group["params"] if p.grad is not None]
        own_ids = {id(p) for p in group["params"]}
        extra = []
        if self._norm_params is not None:
            src = self._norm_params() if callable(self._norm_params) else self._norm_params
            extra = [p for p in src if id(p) not in own_ids and p.grad is not None]
        if not own and not extra:
            return loss
        # validate everything and collect the pointers FIRST; the per-parameter step counters are advanced only after the
        # launch has been accepted, so that a failed step leaves the optimizer where it was and can simply be retried
        steps = set()
        P, G, M, V, N = [], [], [], [], []
        keep = []
        for p in own:
            g = p.grad
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and g.is_cuda and g.dtype == torch.float32):
                raise RuntimeError("ClipAdam: contiguous fp32 CUDA parameters (and fp32 CUDA gradients) only")
            st = self.state[p]
            steps.add(int(st["step"]) + 1 if st else 1)
        for p in extra:
            g = p.grad
            if not (g.is_cuda and g.dtype == torch.float32):
                raise RuntimeError("ClipAdam: the clip-only tensors of norm_params need fp32 CUDA gradients "
                                   f"(got {g.dtype} on {g.device})")
        if len(steps) > 1:
            raise RuntimeError("ClipAdam: parameters are at different Adam steps (a parameter without gradient in an earlier step); "
                               "use torch.optim.Adam for such models")
        step = steps.pop() if steps else 1
        for p in own:
            g = p.grad
            if not g.is_contiguous():
                g = p.grad = g.contiguous()
            st = self.state[p]
            if not st:
                st["step"] = torch.tensor(0.0)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            P.append(p.data_ptr()); G.append(g.data_ptr()); M.append(st["exp_avg"].data_ptr()); V.append(st["exp_avg_sq"].data_ptr())
            N.append(p.numel())
        for p in extra:                                   # clip-only tensors (in the norm, scaled, never updated)
            g = p.grad
            if not g.is_contiguous():
                g = p.grad = g.contiguous()
            P.append(None); G.append(g.data_ptr()); M.append(None); V.append(None); N.append(g.numel())
            keep.append(g)
        n = len(G)
        vp = ctypes.c_void_p * n
        numel = (ctypes.c_long * n)(*N)
        lib = _lib.load()
        dev = (own or extra)[0].device
        nws = lib.wf3d_clip_adam_ws_floats(numel, n)
        ws = torch.empty(nws, dtype=torch.float32, device=dev)
        norm = torch.empty((), dtype=torch.float32, device=dev)
        b1, b2 = group["betas"]
        check(lib.wf3d_clip_adam_step(vp(*P), vp(*G), vp(*M), vp(*V), numel, n, float(self.max_norm or 0.0), float(group["lr"]),
                                      float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]), step,
                                      ws.data_ptr(), nws, norm.data_ptr(), _stream()), "clip_adam_step")
        for p in own:
            self.state[p]["step"] += 1
        # the kernel wrote parameters, gradients and moments through raw pointers: tell autograd's version counters
        # (a graph that still holds the old values must fail loudly, as it does after torch.optim.Adam.step())
        bump = torch.autograd.graph.increment_version
        for p in own:
            bump(p)
            bump(p.grad)
        for g in keep:
            bump(g)
        self.last_grad_norm = norm
        return loss
