"""Host side of csrc/skinny.hip: Linear layers over M <= 32 rows (one per cloud of the batch) — the feature-fusion
MLP (reference models/PointNetEncoder.py:57-65,116) and the vertex head (models/VertexPredictor.py:94-117).

A `Stage` is one Linear (+ optional LayerNorm/activation behind it).  Forward launches write the Linear's output z and
per-16-column (mean, M2) partials of it; the launch that consumes act(LN(z)) merges the partials on load.  Backward
alternates two launches per stage: `bwd` (weight/bias gradient + dgrad partial slabs, for up to 4 Linears that are
ready together) and `reduce` (slab sum + the elementwise half of the next LayerNorm backward)."""
import ctypes
import os

import torch

from . import _lib
from ._lib import SkinnyBwd, SkinnyFwd, SkinnyRed, check
from .ops import ACT_NONE, LN_EPS, _need_cuda, _p, _stream

MAX_ROWS = 32


def ok(M, *mats):
    """The skinny kernels take M <= 32 rows and contiguous fp32 [N, K] weights with K % 4 == 0 (16-byte row loads)."""
    if not (1 <= M <= MAX_ROWS):
        return False
    for W in mats:
        if W.dim() != 2 or W.shape[1] % 4 or W.stride(1) != 1 or W.stride(0) % 4 or W.data_ptr() % 16:
            return False
    return True


class LNIn:
    """Input prologue x' = act(LayerNorm(X; gamma, beta)): statistics from `part` (merged in the kernel, and returned
    as (mu, rs) for backward) or given as (mu, rs)."""
    __slots__ = ("gamma", "beta", "act", "part", "mu", "rs")

    def __init__(self, gamma, beta, act, part=None, mu=None, rs=None):
        self.gamma, self.beta, self.act, self.part, self.mu, self.rs = gamma, beta, act, part, mu, rs


class Fwd:
    """One forward problem: Y = (ln(X) + in_add)·W^T + bias + out_add; stats=True also emits Y's statistics partials."""
    __slots__ = ("X", "W", "bias", "ln", "in_add", "out_add", "stats")

    def __init__(self, X, W, bias, ln=None, in_add=None, out_add=None, stats=False):
        self.X, self.W, self.bias, self.ln, self.in_add, self.out_add, self.stats = X, W, bias, ln, in_add, out_add, stats


def _ld(t):
    return t.stride(0) if t.shape[0] > 1 else t.shape[1]


def fwd(M, *specs):
    """One launch for up to 4 problems.  Returns per problem (Y, part | None, (mu, rs) | None)."""
    arr = (SkinnyFwd * len(specs))()
    outs = []
    for d, s in zip(arr, specs):
        _need_cuda(s.X, s.W, s.bias, s.in_add, s.out_add)
        N, K = s.W.shape
        dev = s.X.device
        Y = torch.empty(M, N, dtype=torch.float32, device=dev)
        d.X, d.ldx, d.W, d.ldw, d.bias = _p(s.X), _ld(s.X), _p(s.W), s.W.stride(0), _p(s.bias)
        d.Y, d.ldy, d.N, d.K = _p(Y), N, N, K
        stats = None
        if s.ln is not None:
            ln = s.ln
            d.gamma, d.beta, d.act = _p(ln.gamma), _p(ln.beta), ln.act
            if ln.part is not None:
                d.in_part, d.in_nblk = _p(ln.part), ln.part.shape[0]
                mu = torch.empty(2, M, dtype=torch.float32, device=dev)
                d.mu_out, d.rs_out = _p(mu[0]), _p(mu[1])
                stats = (mu[0], mu[1])
            else:
                d.mu, d.rs = _p(ln.mu), _p(ln.rs)
                stats = (ln.mu, ln.rs)
        else:
            d.act = ACT_NONE
        if s.in_add is not None:
            d.in_addend, d.ld_in_addend = _p(s.in_add), _ld(s.in_add)
        if s.out_add is not None:
            d.out_addend, d.ld_out_addend = _p(s.out_add), _ld(s.out_add)
        part = None
        if s.stats:
            part = torch.empty(N // 16, M, 2, dtype=torch.float32, device=dev)
            d.stat_part = _p(part)
        outs.append((Y, part, stats))
    check(_lib.load().wf3d_skinny_fwd(arr, len(specs), M, LN_EPS, _stream()), "skinny_fwd")
    return outs


class LNOut:
    """This Linear's output went through act(LayerNorm(.)): its dz is rebuilt from G (reduce()), z, (mu, rs), rowpart."""
    __slots__ = ("z", "mu", "rs", "rowpart")

    def __init__(self, z, mu, rs, rowpart):
        self.z, self.mu, self.rs, self.rowpart = z, mu, rs, rowpart


class Bwd:
    """Backward of one Linear: dY = plain dz, or G with ln_out=LNOut(...).  X (+ x_ln, x_add) is the forward input.
    want_dx=False skips the dgrad slabs."""
    __slots__ = ("dY", "ln_out", "W", "X", "x_ln", "x_add", "want_dx", "want_dw")

    def __init__(self, dY, W, X, ln_out=None, x_ln=None, x_add=None, want_dx=True, want_dw=True):
        self.dY, self.W, self.X, self.ln_out, self.x_ln, self.x_add = dY, W, X, ln_out, x_ln, x_add
        self.want_dx, self.want_dw = want_dx, want_dw


def _pick_nc(N, K):
    """dgrad chunk height (weight rows per workgroup).  A chunk of nc rows costs one slab of M*K*4 bytes (written, read
    back by reduce) and nc/2 x 4 fp32 MFMAs of 64 cycles per wave.  Measured GPU-side over the nine head layers of cfg2
    (rocprofv3, bwd + reduce per step): nc = 32: 305 us, 64: 270 us, 128: 289 us, the earlier mixed rule (128 for the big
    and the narrow layers) 292 us — so 64 everywhere (WF3D_SKINNY_NC forces another value)."""
    force = os.environ.get("WF3D_SKINNY_NC")
    if force:
        return min(int(force), max(2, N + (N & 1)))
    if N < 64:
        return max(2, N + (N & 1))
    return 64


def bwd(M, *specs):
    """One launch.  Returns per problem (dW | None, db | None, slabs | None)."""
    arr = (SkinnyBwd * len(specs))()
    outs = []
    for d, s in zip(arr, specs):
        _need_cuda(s.dY, s.W, s.X, s.x_add)
        N, K = s.W.shape
        dev = s.W.device
        d.dY, d.lddy = _p(s.dY), _ld(s.dY)
        if s.ln_out is not None:
            lo = s.ln_out
            d.z, d.ldz, d.mu, d.rs = _p(lo.z), _ld(lo.z), _p(lo.mu), _p(lo.rs)
            d.rowpart, d.rowpart_nblk = _p(lo.rowpart), lo.rowpart.shape[0]
        d.W, d.ldw, d.N, d.K = _p(s.W), s.W.stride(0), N, K
        dW = db = slabs = None
        if s.want_dw:
            d.X, d.ldx = _p(s.X), _ld(s.X)
            if s.x_ln is not None:
                xl = s.x_ln
                d.xmu, d.xrs, d.xgamma, d.xbeta, d.xact = _p(xl.mu), _p(xl.rs), _p(xl.gamma), _p(xl.beta), xl.act
            if s.x_add is not None:
                d.xadd, d.ldxadd = _p(s.x_add), _ld(s.x_add)
            dW = torch.empty(N, K, dtype=torch.float32, device=dev)
            db = torch.empty(N, dtype=torch.float32, device=dev)
            d.dW, d.lddw, d.db = _p(dW), K, _p(db)
        if s.want_dx:
            nc = _pick_nc(N, K)
            slabs = torch.empty(-(-N // nc), M, K, dtype=torch.float32, device=dev)
            d.slabs, d.nc = _p(slabs), nc
        outs.append((dW, db, slabs))
    check(_lib.load().wf3d_skinny_bwd(arr, len(specs), M, _stream()), "skinny_bwd")
    return outs


def reduce(M, K, slab_sets, extra=None, want_dh=True, ln=None, dev=None):
    """v = sum of the slab sets (+ extra).  ln = (z, mu, rs, gamma, beta, act): also the elementwise LayerNorm/activation
    backward of the stage whose output v is the gradient of.  Returns (dh | None, G, dgamma, dbeta, rowpart) (None's
    without ln)."""
    d = SkinnyRed()
    if len(slab_sets) > 3:
        raise RuntimeError("wf3d.skinny.reduce: at most three slab sets")
    dev = dev or (slab_sets[0].device if slab_sets else extra.device)
    for i, sl in enumerate(slab_sets):
        d.slabs[i], d.nslab[i] = _p(sl), sl.shape[0]
    if extra is not None:
        d.extra, d.ldextra = _p(extra), _ld(extra)
    d.K = K
    dh = None
    if want_dh:
        dh = torch.empty(M, K, dtype=torch.float32, device=dev)
        d.dh, d.lddh = _p(dh), K
    G = dgb = rowpart = None
    if ln is not None:
        z, mu, rs, gamma, beta, act = ln
        d.z, d.ldz, d.mu, d.rs, d.gamma, d.beta, d.act = _p(z), _ld(z), _p(mu), _p(rs), _p(gamma), _p(beta), act
        G = torch.empty(M, K, dtype=torch.float32, device=dev)
        dgb = torch.empty(2, K, dtype=torch.float32, device=dev)
        rowpart = torch.empty(-(-K // 64), M, 2, dtype=torch.float32, device=dev)
        d.G, d.ldg, d.dgamma, d.dbeta, d.rowpart = _p(G), K, _p(dgb[0]), _p(dgb[1]), _p(rowpart)
    check(_lib.load().wf3d_skinny_reduce(ctypes.byref(d), M, _stream()), "skinny_reduce")
    if ln is None:
        return dh, None, None, None, None
    return dh, G, dgb[0], dgb[1], rowpart
