"""Thin torch-tensor wrappers over the C ABI (one Python function per entry
point of include/wf3d.h).  torch is used for device memory and the current
stream only; every FLOP happens in libwf3d.so.  CPU tensors are rejected —
there is no fallback path.
"""
import ctypes

import torch

from . import _lib
from ._lib import GemmDesc, check

ACT_NONE, ACT_RELU, ACT_GELU = 0, 1, 2
NT, NN, TN = 0, 1, 2
LN_EPS = 1e-5

_scratch = {}


def _need_cuda(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("wf3d: the HIP path needs CUDA (ROCm) tensors; there is no CPU fallback")
        if t.dtype not in (torch.float32, torch.int32, torch.int64):
            raise RuntimeError(f"wf3d: unsupported dtype {t.dtype}")


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def scratch(nbytes, device, slot=0):
    """Grow-only per-(device, stream, slot) scratch buffer.  Kernels run in
    stream order, so reuse across consecutive calls on one stream is safe."""
    if nbytes <= 0:
        return None
    key = (device.index, torch.cuda.current_stream().cuda_stream, slot)
    buf = _scratch.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _scratch[key] = buf
    return buf


def _rows2d(t):
    if t.dim() != 2 or t.stride(1) != 1:
        raise RuntimeError("wf3d: expected a 2-D tensor with unit inner stride")
    return t


class Pro:
    """Prologue spec for `gemm`: v' = drop(act(LN-affine(v))) on the activation operand."""
    __slots__ = ("act", "mu", "rs", "gamma", "beta", "drop_p", "seed")

    def __init__(self, act, mu=None, rs=None, gamma=None, beta=None, drop_p=0.0, seed=0):
        self.act, self.mu, self.rs, self.gamma, self.beta = act, mu, rs, gamma, beta
        self.drop_p, self.seed = float(drop_p), int(seed) & 0xFFFFFFFF


def gemm(a, b, layout, bias=None, addend=None, out=None, accumulate=False, pro=None):
    """C = pro(A)·B (+bias) (+addend) (+C).  NT: a[M,K] b[N,K]; NN: a[M,K] b[K,N]; TN: a[K,M] b[K,N]."""
    _need_cuda(a, b, bias, addend, out)
    a, b = _rows2d(a), _rows2d(b)
    if layout == NT:
        M, K = a.shape; N, K2 = b.shape
    elif layout == NN:
        M, K = a.shape; K2, N = b.shape
    elif layout == TN:
        K, M = a.shape; K2, N = b.shape
    else:
        raise ValueError("bad layout")
    if K != K2:
        raise RuntimeError(f"wf3d.gemm: reduction dims differ ({K} vs {K2})")
    if out is None:
        if accumulate:
            raise RuntimeError("wf3d.gemm: accumulate needs `out`")
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    else:
        _rows2d(out)
        if tuple(out.shape) != (M, N):
            raise RuntimeError(f"wf3d.gemm: out shape {tuple(out.shape)} != {(M, N)}")
    if bias is not None and (bias.numel() != N or not bias.is_contiguous()):
        raise RuntimeError("wf3d.gemm: bias must be contiguous [N]")
    if addend is not None:
        _rows2d(addend)
        if tuple(addend.shape) != (M, N):
            raise RuntimeError("wf3d.gemm: addend shape mismatch")
    d = GemmDesc()
    d.A, d.B, d.C = _p(a), _p(b), _p(out)
    d.bias, d.addend = _p(bias), _p(addend)
    d.M, d.N, d.K = M, N, K
    d.lda, d.ldb, d.ldc = a.stride(0) if a.shape[0] > 1 else max(a.shape[1], 1), \
        b.stride(0) if b.shape[0] > 1 else max(b.shape[1], 1), out.stride(0) if M > 1 else max(N, 1)
    d.ld_addend = (addend.stride(0) if M > 1 else N) if addend is not None else 0
    d.layout = layout
    if pro is not None:
        if layout == NN:
            raise RuntimeError("wf3d.gemm: no prologue on NN")
        rows, cols = (M, K) if layout == NT else (K, N)
        for nm, t, n in (("mu", pro.mu, rows), ("rs", pro.rs, rows), ("gamma", pro.gamma, cols), ("beta", pro.beta, cols)):
            if t is not None:
                _need_cuda(t)
                if t.numel() != n or not t.is_contiguous():
                    raise RuntimeError(f"wf3d.gemm: prologue {nm} must be contiguous [{n}]")
        d.pro_enable, d.pro_act = 1, pro.act
        d.pro_mu, d.pro_rs, d.pro_gamma, d.pro_beta = _p(pro.mu), _p(pro.rs), _p(pro.gamma), _p(pro.beta)
        d.drop_p, d.drop_seed = pro.drop_p, pro.seed
    d.accumulate = 1 if accumulate else 0
    lib = _lib.load()
    nb = lib.wf3d_gemm_ws_bytes(M, N, K, layout)
    ws = scratch(nb, a.device)
    d.ws, d.ws_bytes = _p(ws), (ws.numel() if ws is not None else 0)
    if M and N:
        check(lib.wf3d_gemm(ctypes.byref(d), _stream()), "gemm")
    return out


def row_stats(z, eps=LN_EPS):
    _need_cuda(z)
    z = _rows2d(z)
    R, D = z.shape
    mu = torch.empty(R, dtype=torch.float32, device=z.device)
    rs = torch.empty(R, dtype=torch.float32, device=z.device)
    ld = z.stride(0) if R > 1 else D
    check(_lib.load().wf3d_row_stats(_p(z), R, D, ld, eps, _p(mu), _p(rs), _stream()), "row_stats")
    return mu, rs


def ln_act_apply(z, mu, rs, gamma, beta, act, addend=None, drop_p=0.0, seed=0, out=None):
    _need_cuda(z, mu, rs, gamma, beta, addend)
    if not z.is_contiguous() or (addend is not None and not addend.is_contiguous()):
        raise RuntimeError("wf3d.ln_act_apply: contiguous tensors required")
    R, D = z.shape
    if out is None:
        out = torch.empty_like(z)
    check(_lib.load().wf3d_ln_act_apply(_p(z), R, D, _p(mu), _p(rs), _p(gamma), _p(beta), act, _p(addend),
                                        float(drop_p), int(seed) & 0xFFFFFFFF, _p(out), _stream()), "ln_act_apply")
    return out


def ln_act_bwd(dh, z, mu, rs, gamma, beta, act, drop_p=0.0, seed=0, want_affine=True, want_bias=True, inplace=False):
    """Returns (dz, dgamma, dbeta, dbias)."""
    _need_cuda(dh, z, mu, rs, gamma, beta)
    if not (dh.is_contiguous() and z.is_contiguous()):
        raise RuntimeError("wf3d.ln_act_bwd: contiguous tensors required")
    R, D = z.shape
    dz = dh if inplace else torch.empty_like(z)
    dev = z.device
    dgamma = torch.empty(D, dtype=torch.float32, device=dev) if (want_affine and gamma is not None) else None
    dbeta = torch.empty(D, dtype=torch.float32, device=dev) if (want_affine and gamma is not None) else None
    dbias = torch.empty(D, dtype=torch.float32, device=dev) if want_bias else None
    lib = _lib.load()
    ws = scratch(lib.wf3d_ln_act_bwd_ws_bytes(R, D), dev)
    check(lib.wf3d_ln_act_bwd(_p(dh), _p(z), R, D, _p(mu), _p(rs), _p(gamma), _p(beta), act, float(drop_p),
                              int(seed) & 0xFFFFFFFF, _p(dz), _p(dgamma), _p(dbeta), _p(dbias), _p(ws),
                              ws.numel() if ws is not None else 0, _stream()), "ln_act_bwd")
    return dz, dgamma, dbeta, dbias


def colsum(x, w=None):
    _need_cuda(x, w)
    x = _rows2d(x)
    R, D = x.shape
    out = torch.empty(D, dtype=torch.float32, device=x.device)
    lib = _lib.load()
    ws = scratch(lib.wf3d_colsum_ws_bytes(R, D), x.device)
    ld = x.stride(0) if R > 1 else D
    check(lib.wf3d_colsum(_p(x), R, D, ld, _p(w), _p(out), _p(ws), ws.numel() if ws is not None else 0, _stream()),
          "colsum")
    return out


def point_valid(x):
    _need_cuda(x)
    if not x.is_contiguous():
        raise RuntimeError("wf3d.point_valid: contiguous input required")
    M, D = x.numel() // x.shape[-1], x.shape[-1]
    valid = torch.empty(M, dtype=torch.float32, device=x.device)
    check(_lib.load().wf3d_point_valid(_p(x), M, D, _p(valid), _stream()), "point_valid")
    return valid


def pool4_fwd(pf, valid):
    """pf [B,N,C], valid [B*N] -> (mmax, mavg, umean, umax, arg_m, arg_u, cnt)."""
    _need_cuda(pf, valid)
    if not pf.is_contiguous():
        raise RuntimeError("wf3d.pool4_fwd: contiguous point_features required")
    B, N, C = pf.shape
    dev = pf.device
    f = lambda: torch.empty(B, C, dtype=torch.float32, device=dev)   # noqa: E731
    mmax, mavg, umean, umax = f(), f(), f(), f()
    arg_m = torch.empty(B, C, dtype=torch.int32, device=dev)
    arg_u = torch.empty(B, C, dtype=torch.int32, device=dev)
    cnt = torch.empty(B, dtype=torch.float32, device=dev)
    lib = _lib.load()
    ws = scratch(lib.wf3d_pool4_ws_bytes(B, N, C), dev)
    check(lib.wf3d_pool4_fwd(_p(pf), _p(valid), B, N, C, _p(mmax), _p(mavg), _p(umean), _p(umax), _p(arg_m),
                             _p(arg_u), _p(cnt), _p(ws), ws.numel(), _stream()), "pool4_fwd")
    return mmax, mavg, umean, umax, arg_m, arg_u, cnt


def pool4_bwd(valid, cnt, arg_m, arg_u, dmmax, dmavg, dumean, dumax, dpf_direct, B, N, C):
    _need_cuda(valid, cnt, arg_m, arg_u, dmmax, dmavg, dumean, dumax, dpf_direct)
    for t in (dmmax, dmavg, dumean, dumax, dpf_direct):
        if t is not None and not t.is_contiguous():
            raise RuntimeError("wf3d.pool4_bwd: contiguous cotangents required")
    dpf = torch.empty(B, N, C, dtype=torch.float32, device=valid.device)
    check(_lib.load().wf3d_pool4_bwd(_p(valid), _p(cnt), _p(arg_m), _p(arg_u), _p(dmmax), _p(dmavg), _p(dumean),
                                     _p(dumax), _p(dpf_direct), B, N, C, _p(dpf), _stream()), "pool4_bwd")
    return dpf
