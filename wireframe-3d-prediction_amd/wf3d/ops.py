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
    # plain int (or None): every pointer parameter is declared c_void_p in _lib.SIGNATURES, ctypes converts
    return None if t is None else t.data_ptr()


try:                                     # raw handle of the current stream without building a Stream object (~10x cheaper;
    _raw_stream = torch._C._cuda_getCurrentRawStream      # the heads issue ~200 launches of 5-20 us per step: CPU-bound there)
except AttributeError:                   # pragma: no cover
    _raw_stream = None


def _stream_handle():
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _stream():
    return _stream_handle()


def scratch(nbytes, device, slot=0):
    """Grow-only per-(device, stream, slot) scratch buffer.  Kernels run in
    stream order, so reuse across consecutive calls on one stream is safe."""
    if nbytes <= 0:
        return None
    key = (device.index, _stream_handle(), slot)
    buf = _scratch.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _scratch[key] = buf
    return buf


def _rows2d(t):
    if t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
        raise RuntimeError("wf3d: expected a 2-D tensor with unit inner stride")
    return t


class Pro:
    """Prologue spec for `gemm`: v' = drop(act(LN-affine(v))) on the activation operand."""
    __slots__ = ("act", "mu", "rs", "gamma", "beta", "drop_p", "seed")

    def __init__(self, act, mu=None, rs=None, gamma=None, beta=None, drop_p=0.0, seed=0):
        self.act, self.mu, self.rs, self.gamma, self.beta = act, mu, rs, gamma, beta
        self.drop_p, self.seed = float(drop_p), int(seed) & 0xFFFFFFFF


def gemm(a, b, layout, bias=None, addend=None, out=None, accumulate=False, pro=None, x3=False, lowrank=None):
    """C = pro(A)·B (+bias) (+addend) (+C).  NT: a[M,K] b[N,K]; NN: a[M,K] b[K,N]; TN: a[K,M] b[K,N].
    x3=True: bf16x3 arithmetic on the fp32 operands (split while staged; see wf3d_gemm_t.x3); default exact fp32.
    lowrank=(U[M,k], V[N,k]), k <= 4: C += U·V^T in the epilogue (exact fp32; unit inner stride, any row stride)."""
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
    d.x3 = 1 if x3 else 0
    if lowrank is not None:
        u, v = lowrank
        _need_cuda(u, v)
        if u.dim() != 2 or v.dim() != 2 or u.shape[0] != M or v.shape[0] != N or u.shape[1] != v.shape[1] or u.shape[1] > 4 \
                or u.stride(1) != 1 or v.stride(1) != 1:
            raise RuntimeError("wf3d.gemm: lowrank needs U[M,k], V[N,k] with k <= 4 and unit inner stride")
        d.lr_u, d.lr_v, d.lr_k = _p(u), _p(v), u.shape[1]
        d.ld_lr_u, d.ld_lr_v = u.stride(0) if M > 1 else u.shape[1], v.stride(0) if N > 1 else v.shape[1]
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


def ln_act_bwd(dh, z, mu, rs, gamma, beta, act, drop_p=0.0, seed=0, want_affine=True, want_bias=True, inplace=False,
               dz_split=None, want_dz=True):
    """Returns (dz, dgamma, dbeta, dbias); `dz_split` (optional sx8 buffer) also receives dz, and
    with want_dz=False the fp32 dz is not written at all (returned as None)."""
    _need_cuda(dh, z, mu, rs, gamma, beta)
    if not (dh.is_contiguous() and z.is_contiguous()):
        raise RuntimeError("wf3d.ln_act_bwd: contiguous tensors required")
    R, D = z.shape
    if not want_dz and dz_split is None:
        raise RuntimeError("wf3d.ln_act_bwd: want_dz=False needs dz_split")
    dz = None if not want_dz else (dh if inplace else torch.empty_like(z))
    dev = z.device
    if want_affine and gamma is not None and want_bias:
        # one [3, D] buffer -> the C side finalises all three column sums in a single pass
        trio = torch.empty(3, D, dtype=torch.float32, device=dev)
        dgamma, dbeta, dbias = trio[0], trio[1], trio[2]
    else:
        dgamma = torch.empty(D, dtype=torch.float32, device=dev) if (want_affine and gamma is not None) else None
        dbeta = torch.empty(D, dtype=torch.float32, device=dev) if (want_affine and gamma is not None) else None
        dbias = torch.empty(D, dtype=torch.float32, device=dev) if want_bias else None
    lib = _lib.load()
    ws = scratch(lib.wf3d_ln_act_bwd_ws_bytes(R, D), dev)
    check(lib.wf3d_ln_act_bwd(_p(dh), _p(z), R, D, _p(mu), _p(rs), _p(gamma), _p(beta), act, float(drop_p),
                              int(seed) & 0xFFFFFFFF, _p(dz), _p(dz_split), _p(dgamma), _p(dbeta), _p(dbias), _p(ws),
                              ws.numel() if ws is not None else 0, _stream()), "ln_act_bwd")
    return dz, dgamma, dbeta, dbias


def ln_act_bwd_wsum(dh, z, wrow, mu, rs, gamma, beta, act, drop_p=0.0, seed=0, inplace=False):
    """(dz, dgamma, dbeta, wsum) with wsum[c] = sum_r dz[r, c] * wrow[r] from the same pass."""
    _need_cuda(dh, z, wrow, mu, rs, gamma, beta)
    if not (dh.is_contiguous() and z.is_contiguous() and wrow.is_contiguous()):
        raise RuntimeError("wf3d.ln_act_bwd_wsum: contiguous tensors required")
    R, D = z.shape
    dz = dh if inplace else torch.empty_like(z)
    duo = torch.empty(2, D, dtype=torch.float32, device=z.device)
    wsum = torch.empty(D, dtype=torch.float32, device=z.device)
    lib = _lib.load()
    ws = scratch(lib.wf3d_ln_act_bwd_wsum_ws_bytes(R, D), z.device)
    check(lib.wf3d_ln_act_bwd_wsum(_p(dh), _p(z), _p(wrow), R, D, _p(mu), _p(rs), _p(gamma), _p(beta), act, float(drop_p),
                                   int(seed) & 0xFFFFFFFF, _p(dz), None, _p(duo[0]), _p(duo[1]), _p(wsum), _p(ws),
                                   ws.numel(), _stream()), "ln_act_bwd_wsum")
    return dz, duo[0], duo[1], wsum


def colsum(x, w=None, act=ACT_NONE):
    """out[c] = sum_r act(x[r, c]) * w[r]"""
    _need_cuda(x, w)
    x = _rows2d(x)
    R, D = x.shape
    out = torch.empty(D, dtype=torch.float32, device=x.device)
    lib = _lib.load()
    ws = scratch(lib.wf3d_colsum_ws_bytes(R, D), x.device)
    ld = x.stride(0) if R > 1 else D
    check(lib.wf3d_colsum(_p(x), R, D, ld, _p(w), act, _p(out), _p(ws), ws.numel() if ws is not None else 0, _stream()),
          "colsum")
    return out


def rowdot_act_ok(z, W):
    """The row-dot kernels take a one-output Linear on a contiguous [R, D] with D = 8 * 2^k <= 512."""
    return (W.shape[0] == 1 and z.dim() == 2 and z.is_contiguous() and W.is_contiguous()
            and bool(_lib.load().wf3d_rowdot_act_ok(z.shape[1])))


def rowdot_act(z, W, bias, act):
    """[R, 1] = act(z) @ W^T + bias for a one-output Linear W [1, D]."""
    _need_cuda(z, W, bias)
    R, D = z.shape
    out = torch.empty(R, 1, dtype=torch.float32, device=z.device)
    check(_lib.load().wf3d_rowdot_act(_p(z), R, D, _p(W), _p(bias), act, _p(out), _stream()), "rowdot_act")
    return out


def rowdot_act_bwd(z, dlogit, W, act, want_dz=True, dz_split=None):
    """Backward of rowdot_act fused with the activation backward: (dz | None, dW [1, D], dbias_z [D]);
    `dz_split` (optional sx8 buffer) also receives dz."""
    _need_cuda(z, dlogit, W, dz_split)
    R, D = z.shape
    if not want_dz and dz_split is None:
        raise RuntimeError("wf3d.rowdot_act_bwd: want_dz=False needs dz_split")
    dz = torch.empty_like(z) if want_dz else None
    duo = torch.empty(2, D, dtype=torch.float32, device=z.device)
    lib = _lib.load()
    ws = scratch(lib.wf3d_rowdot_act_bwd_ws_bytes(R, D), z.device)
    check(lib.wf3d_rowdot_act_bwd(_p(z), _p(dlogit), R, D, _p(W), act, _p(dz), _p(dz_split), _p(duo[0]), _p(duo[1]),
                                  _p(ws), ws.numel(), _stream()), "rowdot_act_bwd")
    return dz, duo[0].view(1, D), duo[1]


def point_valid(x):
    _need_cuda(x)
    if not x.is_contiguous():
        raise RuntimeError("wf3d.point_valid: contiguous input required")
    M, D = x.numel() // x.shape[-1], x.shape[-1]
    valid = torch.empty(M, dtype=torch.float32, device=x.device)
    check(_lib.load().wf3d_point_valid(_p(x), M, D, _p(valid), _stream()), "point_valid")
    return valid


class PoolOut:
    """Everything one pooling pass yields.  `masked` = [masked max | masked mean] and `unmasked` = [mean | max] are
    [B, 2C] tensors in the order the reference concatenates them (PointNetEncoder.py:115, VertexPredictor.py:88);
    mmax/mavg/umean/umax are views of their halves."""
    __slots__ = ("masked", "unmasked", "mmax", "mavg", "umean", "umax", "arg_m", "arg_u", "cnt", "nvalid")


def pool4_fwd(pf, valid, packed=False):
    """pf [B,N,C], valid [B*N] -> (mmax, mavg, umean, umax, arg_m, arg_u, cnt), or a PoolOut with packed=True."""
    _need_cuda(pf, valid)
    if not pf.is_contiguous():
        raise RuntimeError("wf3d.pool4_fwd: contiguous point_features required")
    B, N, C = pf.shape
    dev = pf.device
    o = PoolOut()
    o.masked = torch.empty(B, 2 * C, dtype=torch.float32, device=dev)
    o.unmasked = torch.empty(B, 2 * C, dtype=torch.float32, device=dev)
    o.mmax, o.mavg = o.masked[:, :C], o.masked[:, C:]
    o.umean, o.umax = o.unmasked[:, :C], o.unmasked[:, C:]
    o.arg_m = torch.empty(B, C, dtype=torch.int32, device=dev)
    o.arg_u = torch.empty(B, C, dtype=torch.int32, device=dev)
    cn = torch.empty(2, B, dtype=torch.float32, device=dev)
    o.cnt, o.nvalid = cn[0], cn[1]
    lib = _lib.load()
    ws = scratch(lib.wf3d_pool4_ws_bytes(B, N, C), dev)
    check(lib.wf3d_pool4_fwd(_p(pf), _p(valid), B, N, C, _p(o.mmax), _p(o.mavg), _p(o.umean), _p(o.umax), 2 * C,
                             _p(o.arg_m), _p(o.arg_u), _p(o.cnt), _p(o.nvalid), _p(ws), ws.numel(), _stream()), "pool4_fwd")
    if packed:
        return o
    return o.mmax, o.mavg, o.umean, o.umax, o.arg_m, o.arg_u, o.cnt


def _pair_ld(a, b, what):
    """Row stride shared by a pair of [B, C] cotangents (the two halves of one [B, 2C] gradient, or contiguous)."""
    ld = None
    for t in (a, b):
        if t is None:
            continue
        if t.dim() != 2 or t.stride(1) != 1:
            raise RuntimeError(f"wf3d.pool4_bwd: {what} cotangents need unit inner stride")
        s = t.stride(0) if t.shape[0] > 1 else t.shape[1]
        if ld is not None and s != ld:
            raise RuntimeError(f"wf3d.pool4_bwd: the two {what} cotangents must share one row stride")
        ld = s
    return ld or 0


def pool4_bwd(valid, cnt, arg_m, arg_u, dmmax, dmavg, dumean, dumax, dpf_direct, B, N, C, sx8=False):
    """sx8=True: the result is written as an sx8 split operand (C % 8 == 0) instead of fp32.  The cotangents may be
    strided views (halves of a [B, 2C] gradient)."""
    _need_cuda(valid, cnt, arg_m, arg_u, dmmax, dmavg, dumean, dumax, dpf_direct)
    if dpf_direct is not None and not dpf_direct.is_contiguous():
        raise RuntimeError("wf3d.pool4_bwd: contiguous dpf_direct required")
    ldm, ldu = _pair_ld(dmmax, dmavg, "masked"), _pair_ld(dumean, dumax, "unmasked")
    dpf = torch.empty(B, N, C, dtype=torch.float32, device=valid.device)
    fn = _lib.load().wf3d_pool4_bwd_sx8 if sx8 else _lib.load().wf3d_pool4_bwd
    check(fn(_p(valid), _p(cnt), _p(arg_m), _p(arg_u), _p(dmmax), _p(dmavg), ldm, _p(dumean), _p(dumax), ldu,
             _p(dpf_direct), B, N, C, _p(dpf), _stream()), "pool4_bwd")
    return dpf


def pool4_bwd_colsum(nvalid, cnt, arg_m, arg_u, dmmax, dmavg, dumean, dumax, dpf_direct, B, N, C):
    """Column sums over all B*N rows of what pool4_bwd writes, from the [B, C] cotangents alone:
    sum_n of  valid*dmavg/cnt + dumean/N + [n==arg_m]*dmmax + [n==arg_u]*dumax  (+ dpf_direct).
    nvalid [B] = number of valid points per cloud (PoolOut.nvalid)."""
    _need_cuda(cnt, arg_m, arg_u, dmmax, dmavg, dumean, dumax, dpf_direct, nvalid)
    ldm, ldu = _pair_ld(dmmax, dmavg, "masked"), _pair_ld(dumean, dumax, "unmasked")
    out = torch.empty(C, dtype=torch.float32, device=cnt.device)
    check(_lib.load().wf3d_pool4_bwd_bias(_p(cnt), _p(nvalid), _p(arg_m), _p(arg_u), _p(dmmax), _p(dmavg), ldm, _p(dumean),
                                          _p(dumax), ldu, B, C, _p(out), _stream()), "pool4_bwd_bias")
    if dpf_direct is not None:              # point_features itself received a gradient (never on the model's own path)
        out += colsum(dpf_direct.view(B * N, C))
    return out


# ---------------------------------------------------------------------------
# vertex-head tail and edge head
# ---------------------------------------------------------------------------
def vertex_finalize_fwd(o, V, vd):
    _need_cuda(o)
    B = o.shape[0]
    exist = torch.empty(B, V, dtype=torch.float32, device=o.device)
    counts = torch.empty(B, dtype=torch.int64, device=o.device)
    check(_lib.load().wf3d_vertex_finalize_fwd(_p(o), B, V, vd, _p(exist), _p(counts), _stream()), "vertex_finalize_fwd")
    return exist, counts


def vertex_finalize_bwd(exist, dexist, d_o_in, B, V, vd):
    """d_o_in: gradient of o [B, V, vd] or of the `vertices` view o[:, :, :3] ([B, V, 3]), contiguous, or None."""
    _need_cuda(exist, dexist, d_o_in)
    for t in (dexist, d_o_in):
        if t is not None and not t.is_contiguous():
            raise RuntimeError("wf3d.vertex_finalize_bwd: contiguous cotangents required")
    in_dim = d_o_in.shape[-1] if d_o_in is not None else vd
    d_o = torch.empty(B, V * vd, dtype=torch.float32, device=exist.device)
    check(_lib.load().wf3d_vertex_finalize_bwd(_p(exist), _p(dexist), _p(d_o_in), in_dim, B, V, vd, _p(d_o), _stream()),
          "vertex_finalize_bwd")
    return d_o


class EdgeMeta:
    """Compact ragged row layout of one batch: sample s owns vertex rows
    voff[s]..voff[s+1]-1 and edge rows eoff[s]..eoff[s+1]-1."""
    _cache = {}

    def __init__(self, counts, device):
        self.counts = tuple(int(c) for c in counts)
        self.B = len(self.counts)
        ecount = [c * (c - 1) // 2 for c in self.counts]
        voff, eoff = [0], [0]
        for c, e in zip(self.counts, ecount):
            voff.append(voff[-1] + c)
            eoff.append(eoff[-1] + e)
        self.Rv, self.Re = voff[-1], eoff[-1]
        self.vmax = max(self.counts, default=0)
        self.max_e = max(ecount, default=0)
        i32 = dict(dtype=torch.int32, device=device)
        self.voff = torch.tensor(voff, **i32)
        self.eoff = torch.tensor(eoff, **i32)
        sid = torch.arange(self.B, **i32)
        self.vsample = torch.repeat_interleave(sid, torch.tensor(self.counts, dtype=torch.int64, device=device))
        self.esample = torch.repeat_interleave(sid, torch.tensor(ecount, dtype=torch.int64, device=device))

    @classmethod
    def get(cls, counts, device):
        key = (device.index, tuple(int(c) for c in counts))
        m = cls._cache.get(key)
        if m is None:
            if len(cls._cache) > 64:
                cls._cache.clear()
            m = cls._cache[key] = cls(counts, device)
        return m


def edge_gather_verts(verts, meta):
    """verts [B, V, d] (any strides with unit inner stride; d = vertex_dim, 1..8) -> cv [Rv, d]."""
    _need_cuda(verts)
    if verts.stride(2) != 1:
        raise RuntimeError("wf3d.edge_gather_verts: inner stride must be 1")
    d = verts.shape[2]
    cv = torch.empty(meta.Rv, d, dtype=torch.float32, device=verts.device)
    check(_lib.load().wf3d_edge_gather_verts(_p(verts), verts.stride(0), verts.stride(1), _p(meta.voff),
                                             _p(meta.vsample), meta.Rv, d, _p(cv), _stream()), "edge_gather_verts")
    return cv


def edge_scatter_dverts(dcv, meta, B, V):
    d = dcv.shape[1]
    out = torch.empty(B, V, d, dtype=torch.float32, device=dcv.device)
    check(_lib.load().wf3d_edge_scatter_dverts(_p(dcv), _p(meta.voff), B, V, d, _p(out), _stream()), "edge_scatter_dverts")
    return out


def attn_fwd(qkv, meta, E, heads, drop_p=0.0, seed=0):
    _need_cuda(qkv)
    ctx = torch.empty(meta.Rv, E, dtype=torch.float32, device=qkv.device)
    lse = torch.empty(meta.Rv, heads, dtype=torch.float32, device=qkv.device)
    check(_lib.load().wf3d_attn_fwd(_p(qkv), _p(meta.voff), meta.B, meta.vmax, E, heads, float(drop_p),
                                    int(seed) & 0xFFFFFFFF, _p(ctx), _p(lse), _stream()), "attn_fwd")
    return ctx, lse


def attn_bwd(qkv, dctx, ctx, lse, meta, E, heads, drop_p=0.0, seed=0):
    _need_cuda(qkv, dctx, ctx, lse)
    if not (dctx.is_contiguous() and ctx.is_contiguous()):
        raise RuntimeError("wf3d.attn_bwd: contiguous dctx / ctx required")
    dqkv = torch.empty_like(qkv)
    check(_lib.load().wf3d_attn_bwd(_p(qkv), _p(dctx), _p(ctx), _p(lse), _p(meta.voff), meta.B, meta.vmax, E, heads,
                                    float(drop_p), int(seed) & 0xFFFFFFFF, _p(dqkv), _stream()), "attn_bwd")
    return dqkv


_wdelta_cache = [None, -1, None]       # (weight tensor, its version, contiguous distance column)


def _wdelta(W0, H):
    """Distance column (the last one) of edge_mlp.0.weight [H, 2H + 2d + 1], gathered once into a contiguous vector (a
    strided per-lane gather of it inside the pair kernels costs 512 uncoalesced loads per edge row).
    Forward and backward of one step see the same weight version and share the copy."""
    if W0.dim() != 2 or W0.shape[0] != H or W0.shape[1] < 2 * H + 3 or (W0.shape[1] - 2 * H - 1) % 2:
        raise RuntimeError("wf3d: edge_mlp.0.weight must be [H, 2H + 2*vertex_dim + 1]")
    c = _wdelta_cache
    if c[0] is W0 and c[1] == W0._version:
        return c[2], 1
    col = W0.detach()[:, W0.shape[1] - 1].contiguous()
    c[0], c[1], c[2] = W0, W0._version, col
    return col, 1


def edge_pair_fwd(Pa, Pb, cv, W0, meta, eps=LN_EPS, ln=None, keep_pre=True):
    """ln = (gamma, beta, act, drop_p, seed): also return h = drop(act(LayerNorm(pre))) as an sx8 operand.
    keep_pre=False (with ln): the pre-activation is not stored (returned as None); edge_pair_ln_bwd rebuilds it."""
    _need_cuda(Pa, Pb, cv, W0)
    H = Pa.shape[1]
    wd, stride = _wdelta(W0, H)
    dev = Pa.device
    pre = torch.empty(meta.Re, H, dtype=torch.float32, device=dev) if (keep_pre or ln is None) else None
    mu = torch.empty(meta.Re, dtype=torch.float32, device=dev)
    rs = torch.empty(meta.Re, dtype=torch.float32, device=dev)
    delta = torch.empty(meta.Re, dtype=torch.float32, device=dev)
    if ln is None:
        check(_lib.load().wf3d_edge_pair_fwd(_p(Pa), _p(Pb), _p(cv), _p(wd), stride, _p(meta.voff), _p(meta.eoff),
                                             _p(meta.esample), meta.Re, H, cv.shape[1], eps, _p(pre), _p(mu), _p(rs), _p(delta),
                                             _stream()), "edge_pair_fwd")
        return pre, mu, rs, delta
    gamma, beta, act, drop_p, seed = ln
    _need_cuda(gamma, beta)
    h = torch.empty(meta.Re, H, dtype=torch.float32, device=dev)
    check(_lib.load().wf3d_edge_pair_fwd_ln(_p(Pa), _p(Pb), _p(cv), _p(wd), stride, _p(meta.voff), _p(meta.eoff),
                                            _p(meta.esample), meta.Re, H, cv.shape[1], eps, _p(pre), _p(mu), _p(rs), _p(delta),
                                            _p(gamma), _p(beta), act, float(drop_p), int(seed) & 0xFFFFFFFF, _p(h),
                                            _stream()), "edge_pair_fwd_ln")
    return pre, mu, rs, delta, h


def edge_pair_ln_bwd(dh, Pa, Pb, delta, W0, meta, mu, rs, gamma, beta, act, drop_p=0.0, seed=0):
    """LayerNorm / activation backward of the first edge layer whose pre-activation was not stored: rebuilt from Pa / Pb.
    Returns (dpre (in place of dh), dgamma, dbeta, wsum) like ln_act_bwd_wsum(..., inplace=True)."""
    _need_cuda(dh, Pa, Pb, delta, W0, mu, rs, gamma, beta)
    if not (dh.is_contiguous() and Pa.is_contiguous() and Pb.is_contiguous() and delta.is_contiguous()):
        raise RuntimeError("wf3d.edge_pair_ln_bwd: contiguous tensors required")
    Re, H = dh.shape
    wd, stride = _wdelta(W0, H)
    duo = torch.empty(2, H, dtype=torch.float32, device=dh.device)
    wsum = torch.empty(H, dtype=torch.float32, device=dh.device)
    lib = _lib.load()
    ws = scratch(lib.wf3d_edge_pair_ln_bwd_ws_bytes(Re, H), dh.device)
    check(lib.wf3d_edge_pair_ln_bwd(_p(dh), _p(Pa), _p(Pb), _p(delta), _p(wd), stride, _p(meta.voff), _p(meta.eoff),
                                    _p(meta.esample), Re, H, _p(mu), _p(rs), _p(gamma), _p(beta), act, float(drop_p),
                                    int(seed) & 0xFFFFFFFF, _p(dh), _p(duo[0]), _p(duo[1]), _p(wsum), _p(ws), ws.numel(),
                                    _stream()), "edge_pair_ln_bwd")
    return dh, duo[0], duo[1], wsum


def edge_pair_bwd(dpre, delta, cv, W0, meta, coord=False):
    """coord=True: dcv also receives dPa·Wc + dPb·Wd (the coordinate columns 2H..2H+2d-1 of W0, d = cv.shape[1])."""
    _need_cuda(dpre, delta, cv, W0)
    H = dpre.shape[1]
    d = cv.shape[1]
    wd, stride = _wdelta(W0, H)
    wc = W0[:, 2 * H:2 * H + 2 * d] if coord else None
    if coord and (W0.dim() != 2 or W0.shape[1] < 2 * H + 2 * d or W0.stride(1) != 1):
        raise RuntimeError("wf3d.edge_pair_bwd: W0 must be [H, >= 2H + 2*vertex_dim] with unit inner stride")
    dev = dpre.device
    dPa = torch.empty(meta.Rv, H, dtype=torch.float32, device=dev)
    dPb = torch.empty(meta.Rv, H, dtype=torch.float32, device=dev)
    dcv = torch.empty(meta.Rv, d, dtype=torch.float32, device=dev)
    check(_lib.load().wf3d_edge_pair_bwd(_p(dpre), _p(delta), _p(cv), _p(wd), stride, _p(meta.voff), _p(meta.eoff),
                                         _p(meta.vsample), meta.Rv, H, d, _p(dPa), _p(dPb), _p(dcv), _p(wc),
                                         W0.stride(0) if coord else 0, _stream()),
          "edge_pair_bwd")
    return dPa, dPb, dcv


def edge_prob_fwd(logit, meta):
    _need_cuda(logit)
    probs = torch.empty(meta.B, meta.max_e, dtype=torch.float32, device=logit.device)   # the kernel writes the 0.0 padding too
    check(_lib.load().wf3d_edge_prob_fwd(_p(logit), _p(meta.eoff), meta.B, meta.max_e, _p(probs), _stream()), "edge_prob_fwd")
    return probs


def edge_prob_bwd(probs, dprobs, meta):
    _need_cuda(probs, dprobs)
    if not dprobs.is_contiguous():
        raise RuntimeError("wf3d.edge_prob_bwd: contiguous dprobs required")
    dlogit = torch.empty(meta.Re, 1, dtype=torch.float32, device=probs.device)
    check(_lib.load().wf3d_edge_prob_bwd(_p(probs), _p(dprobs), _p(meta.eoff), _p(meta.esample), meta.Re, meta.max_e,
                                         _p(dlogit), _stream()), "edge_prob_bwd")
    return dlogit


# ---------------------------------------------------------------------------
# bf16x3 split-precision path ("sx8" operands, see include/wf3d.h)
# ---------------------------------------------------------------------------
def split_rows(t, transpose=False):
    """fp32 [R, C] -> sx8 tensor of the same shape (C % 8 == 0); transpose=True gives split(t.T)."""
    _need_cuda(t)
    t = _rows2d(t)
    if transpose:
        R, C, rs, cs = t.shape[1], t.shape[0], 1, t.stride(0)
    else:
        R, C, rs, cs = t.shape[0], t.shape[1], t.stride(0), 1
    out = torch.empty(R, C, dtype=torch.float32, device=t.device)
    check(_lib.load().wf3d_split_rows(_p(t), rs, cs, R, C, _p(out), _stream()), "split_rows")
    return out


def split_weights(weights):
    """[(split(W), split(W^T)) for W in weights] — both sx8 orientations of up to 4 weight matrices per launch
    (the forward GEMM's operand and its dgrad's)."""
    out = []
    for i in range(0, len(weights), 4):
        chunk = weights[i:i + 4]
        n = 2 * len(chunk)
        ins, outs = (ctypes.c_void_p * n)(), (ctypes.c_void_p * n)()
        rs, cs = (ctypes.c_long * n)(), (ctypes.c_long * n)()
        Rr, Cc = (ctypes.c_int * n)(), (ctypes.c_int * n)()
        res = []
        for k, W in enumerate(chunk):
            _need_cuda(W)
            W = _rows2d(W)
            N, K = W.shape
            a = torch.empty(N, K, dtype=torch.float32, device=W.device)
            b = torch.empty(K, N, dtype=torch.float32, device=W.device)
            ins[2 * k], outs[2 * k], rs[2 * k], cs[2 * k], Rr[2 * k], Cc[2 * k] = W.data_ptr(), a.data_ptr(), W.stride(0), 1, N, K
            ins[2 * k + 1], outs[2 * k + 1], rs[2 * k + 1], cs[2 * k + 1], Rr[2 * k + 1], Cc[2 * k + 1] = \
                W.data_ptr(), b.data_ptr(), 1, W.stride(0), K, N
            res.append((a, b))
        check(_lib.load().wf3d_split_rows_multi(ins, rs, cs, Rr, Cc, outs, n, _stream()), "split_rows_multi")
        out += res
    return out


def ln_prep(z, gamma, beta, act, eps=LN_EPS, drop_p=0.0, seed=0):
    """(mu, rs, h_sx8) with h = drop(act(LayerNorm(z))); z contiguous [R, D], D % 8 == 0."""
    _need_cuda(z, gamma, beta)
    if not z.is_contiguous():
        raise RuntimeError("wf3d.ln_prep: contiguous z required")
    R, D = z.shape
    mu = torch.empty(R, dtype=torch.float32, device=z.device)
    rs = torch.empty(R, dtype=torch.float32, device=z.device)
    h = torch.empty_like(z)
    check(_lib.load().wf3d_ln_prep(_p(z), R, D, _p(gamma), _p(beta), act, eps, float(drop_p), int(seed) & 0xFFFFFFFF,
                                   _p(mu), _p(rs), _p(h), _stream()), "ln_prep")
    return mu, rs, h


def first_layer_ok(x, W):
    """Shapes the fused first-layer kernels take: in_features <= 8, out_features a multiple of 8 up to 1024."""
    return x.dim() == 2 and x.shape[1] <= 8 and W.shape[0] % 8 == 0 and W.shape[0] <= 1024 and x.stride(1) == 1 and W.stride(1) == 1


def first_layer_fwd(x, W, bias, gamma, beta, act, eps=LN_EPS):
    """x [R, K<=8] -> (z = x·W^T + b, mu, rs, h_sx8 = act(LayerNorm(z))) in one pass."""
    _need_cuda(x, W, bias, gamma, beta)
    R, K = x.shape
    D = W.shape[0]
    z = torch.empty(R, D, dtype=torch.float32, device=x.device)
    h = torch.empty_like(z)
    mu = torch.empty(R, dtype=torch.float32, device=x.device)
    rs = torch.empty(R, dtype=torch.float32, device=x.device)
    check(_lib.load().wf3d_first_layer_fwd(_p(x), R, K, x.stride(0), _p(W), W.stride(0), _p(bias), D, _p(gamma), _p(beta),
                                           act, eps, _p(z), _p(mu), _p(rs), _p(h), _stream()), "first_layer_fwd")
    return z, mu, rs, h


def ln_act_bwd_first(dh, z, x, mu, rs, gamma, beta, act):
    """Backward of act(LayerNorm(z)) for the first layer: (dgamma, dbeta, dbias, dW[D, K]); dz is not materialised."""
    _need_cuda(dh, z, x, mu, rs, gamma, beta)
    if not (dh.is_contiguous() and z.is_contiguous()):
        raise RuntimeError("wf3d.ln_act_bwd_first: contiguous tensors required")
    R, D = z.shape
    K = x.shape[1]
    trio = torch.empty(3, D, dtype=torch.float32, device=z.device)
    dW = torch.empty(D, K, dtype=torch.float32, device=z.device)
    lib = _lib.load()
    ws = scratch(lib.wf3d_ln_act_bwd_first_ws_bytes(R, D), z.device)
    check(lib.wf3d_ln_act_bwd_first(_p(dh), _p(z), _p(x), R, D, K, x.stride(0), _p(mu), _p(rs), _p(gamma), _p(beta), act,
                                    _p(trio[0]), _p(trio[1]), _p(trio[2]), _p(dW), _p(ws), ws.numel(), _stream()),
          "ln_act_bwd_first")
    return trio[0], trio[1], trio[2], dW


def gemm_split(a_s, b_s, bias=None, out=None, accumulate=False):
    """C[M,N] = A·B^T (+bias) with A = sx8[M,K], B = sx8[N,K] (bf16x3: hi*hi + hi*lo + lo*hi, fp32 accumulate)."""
    _need_cuda(a_s, b_s, bias, out)
    a_s, b_s = _rows2d(a_s), _rows2d(b_s)
    M, K = a_s.shape
    N, K2 = b_s.shape
    if K != K2:
        raise RuntimeError("wf3d.gemm_split: reduction dims differ")
    if out is None:
        if accumulate:
            raise RuntimeError("wf3d.gemm_split: accumulate needs `out`")
        out = torch.empty(M, N, dtype=torch.float32, device=a_s.device)
    lib = _lib.load()
    ws = scratch(lib.wf3d_gemm_split_ws_bytes(M, N, K), a_s.device)
    check(lib.wf3d_gemm_split(_p(a_s), _p(b_s), _p(out), _p(bias), M, N, K, a_s.stride(0), b_s.stride(0),
                              out.stride(0), 1 if accumulate else 0, _p(ws), ws.numel() if ws is not None else 0,
                              _stream()), "gemm_split")
    return out


def split_transpose(t, pro=None, in_sx8=False):
    """sx8 [C, R] = split(pro(t)^T) for fp32 t [R, C] (R % 8 == 0); pro = Pro(act, mu, rs, gamma, beta) or None.
    in_sx8=True: t is itself an sx8 matrix and is transposed plane-wise (no prologue)."""
    _need_cuda(t)
    t = _rows2d(t)
    R, C = t.shape
    out = torch.empty(C, R, dtype=torch.float32, device=t.device)
    mu = rs = gamma = beta = None
    act, drop_p, seed = ACT_NONE, 0.0, 0
    if pro is not None:
        mu, rs, gamma, beta, act, drop_p, seed = pro.mu, pro.rs, pro.gamma, pro.beta, pro.act, pro.drop_p, pro.seed
    check(_lib.load().wf3d_split_transpose(_p(t), R, C, t.stride(0), _p(mu), _p(rs), _p(gamma), _p(beta), act,
                                           float(drop_p), int(seed) & 0xFFFFFFFF, 1 if in_sx8 else 0, _p(out),
                                           _stream()), "split_transpose")
    return out


def gemm_split_tn_ok(a_s, b_s):
    return bool(_lib.load().wf3d_gemm_split_tn_ok(a_s.shape[1], b_s.shape[1], a_s.shape[0], a_s.stride(0), b_s.stride(0)))


def gemm_split_tn_shape_ok(K, Mo, No, lda, ldb):
    """Same test from shapes alone (before the operand exists)."""
    return bool(_lib.load().wf3d_gemm_split_tn_ok(Mo, No, K, lda, ldb))


def gemm_split_tn(a_s, b_s, out=None, accumulate=False):
    """C[Mo,No] = A^T·B with A = sx8[K,Mo], B = sx8[K,No] (wgrad on operands as stored)."""
    _need_cuda(a_s, b_s, out)
    a_s, b_s = _rows2d(a_s), _rows2d(b_s)
    K, Mo = a_s.shape
    K2, No = b_s.shape
    if K != K2:
        raise RuntimeError("wf3d.gemm_split_tn: reduction dims differ")
    if out is None:
        if accumulate:
            raise RuntimeError("wf3d.gemm_split_tn: accumulate needs `out`")
        out = torch.empty(Mo, No, dtype=torch.float32, device=a_s.device)
    lib = _lib.load()
    ws = scratch(lib.wf3d_gemm_split_tn_ws_bytes(Mo, No, K), a_s.device)
    check(lib.wf3d_gemm_split_tn(_p(a_s), _p(b_s), _p(out), Mo, No, K, a_s.stride(0), b_s.stride(0), out.stride(0),
                                 1 if accumulate else 0, _p(ws), ws.numel() if ws is not None else 0, _stream()),
          "gemm_split_tn")
    return out


# ---------------------------------------------------------------------------
# row f-1: WireframeLoss on device
# ---------------------------------------------------------------------------
def loss_cost_matrix(verts, exist, tverts, counts):
    """verts [B,V,3] (view ok), exist [B,V], tverts [B,Vt,3], counts int64 [B] -> cost [B,V,V]."""
    _need_cuda(verts, exist, tverts, counts)
    if verts.stride(2) != 1 or not exist.is_contiguous() or not tverts.is_contiguous() or counts.dtype != torch.int64:
        raise RuntimeError("wf3d.loss_cost_matrix: bad layout / dtype")
    B, V, _ = verts.shape
    cost = torch.empty(B, V, V, dtype=torch.float32, device=verts.device)
    check(_lib.load().wf3d_loss_cost_matrix(_p(verts), verts.stride(0), verts.stride(1), _p(exist), _p(tverts),
                                            tverts.shape[1], _p(counts), B, V, _p(cost), _stream()), "loss_cost_matrix")
    return cost


def loss_terms(verts, exist, edge, tverts, texist, tlabel, m_pred, m_tgt, m_off, n_match, weights):
    """Returns (losses[4] = vertex, existence, edge, total; dverts [B,V,3]; dexist [B,V]; dedge [B,Ep])."""
    _need_cuda(verts, exist, edge, tverts, texist, tlabel, m_pred, m_tgt, m_off)
    B, V, _ = verts.shape
    Ep = edge.shape[1] if edge is not None else 0
    Et = tlabel.shape[1] if tlabel is not None else 0
    dev = verts.device
    dverts = torch.empty(B, V, 3, dtype=torch.float32, device=dev)
    dexist = torch.empty(B, V, dtype=torch.float32, device=dev)
    dedge = torch.empty(B, Ep, dtype=torch.float32, device=dev)
    losses = torch.empty(4, dtype=torch.float32, device=dev)
    ws = scratch(B * 3 * 4, dev)
    check(_lib.load().wf3d_loss_terms(_p(verts), verts.stride(0), verts.stride(1), _p(exist), _p(edge), Ep, _p(tverts),
                                      tverts.shape[1], _p(texist), _p(tlabel), Et, _p(m_pred), _p(m_tgt), _p(m_off),
                                      int(n_match), B, V, float(weights[0]), float(weights[1]), float(weights[2]),
                                      _p(dverts), _p(dexist), _p(dedge), _p(losses), _p(ws), ws.numel(), _stream()),
          "loss_terms")
    return losses, dverts, dexist, dedge


def loss_assign(cost, counts=None):
    """cost [B,V,V] -> col4row int32 [B,V]: optimal assignment per sample, computed on the device.  With `counts`
    (int64 [B]; columns >= counts[b] are the wireframe cost's identical dummy columns) only the real targets are
    augmented: same matches to real targets, unmatched predictions on the dummy columns in order."""
    _need_cuda(cost, counts)
    B, V, _ = cost.shape
    out = torch.empty(B, V, dtype=torch.int32, device=cost.device)
    if counts is None:
        check(_lib.load().wf3d_loss_assign(_p(cost), B, V, _p(out), _stream()), "loss_assign")
    else:
        check(_lib.load().wf3d_loss_assign_counts(_p(cost), _p(counts), B, V, _p(out), _stream()), "loss_assign_counts")
    return out


def loss_terms_assigned(verts, exist, edge, tverts, texist, tlabel, col4row, counts, weights):
    _need_cuda(verts, exist, edge, tverts, texist, tlabel, col4row, counts)
    B, V, _ = verts.shape
    Ep = edge.shape[1] if edge is not None else 0
    Et = tlabel.shape[1] if tlabel is not None else 0
    dev = verts.device
    dverts = torch.empty(B, V, 3, dtype=torch.float32, device=dev)
    dexist = torch.empty(B, V, dtype=torch.float32, device=dev)
    dedge = torch.empty(B, Ep, dtype=torch.float32, device=dev)
    losses = torch.empty(4, dtype=torch.float32, device=dev)
    ws = scratch(B * 3 * 4, dev)
    check(_lib.load().wf3d_loss_terms_assigned(_p(verts), verts.stride(0), verts.stride(1), _p(exist), _p(edge), Ep,
                                               _p(tverts), tverts.shape[1], _p(texist), _p(tlabel), Et, _p(col4row),
                                               _p(counts), B, V, float(weights[0]), float(weights[1]), float(weights[2]),
                                               _p(dverts), _p(dexist), _p(dedge), _p(losses), _p(ws), ws.numel(),
                                               _stream()), "loss_terms_assigned")
    return losses, dverts, dexist, dedge
