"""The three stages of the hot path as torch.autograd.Function's whose forward
and backward are sequences of libwf3d.so calls (wf3d.ops).

Stage          reference lines replaced
EncoderFn      models/PointNetEncoder.py:67-118
VertexFn       models/VertexPredictor.py:63-133
EdgeFn         models/EdgePredictor.py:91-140 x the per-sample loop of
               models/PointCloudToWireframe.py:77-112 (batched, ragged)

Data layout in HBM (fp32): per Linear only its PRE-LayerNorm output z and the
per-row (mu, rstd) are stored; LN-apply + ReLU/GELU (+dropout) are re-applied in
the staging pass of whichever GEMM consumes them (forward GEMM, wgrad GEMM), so
normalised activations never exist in memory.
"""
import torch

from . import config, ops
from . import skinny as sk
from .ops import ACT_GELU, ACT_NONE, ACT_RELU, NN, NT, TN, Pro


def _saved(ctx):
    """The activations a stage kept for its backward.  They are released by the first backward (2.7 GB
    at cfg2), so a second one through the same graph (retain_graph=True) cannot be served: say so
    instead of failing on a None unpack."""
    if ctx.saved is None:
        raise RuntimeError("wf3d: backward through this graph a second time — the stage's saved activations were "
                           "released by the first backward; re-run the forward, or set wf3d.config.RETAIN_SAVED = True "
                           "(WF3D_RETAIN_SAVED=1) before the forward to keep them for backward(retain_graph=True)")
    return ctx.saved


def _keep_params(ctx, params):
    """Parameters are kept as plain attributes (save_for_backward would forbid the views the stages take of them);
    their version counters are recorded so that an in-place update between forward and backward is reported the way
    autograd reports it for saved tensors, instead of silently producing gradients against the new values."""
    ctx.params = params
    ctx.param_versions = [None if p is None else p._version for p in params]


def _params(ctx):
    for i, (p, v) in enumerate(zip(ctx.params, ctx.param_versions)):
        if p is not None and p._version != v:
            raise RuntimeError(f"wf3d: parameter {i} of this stage (shape {tuple(p.shape)}) was modified in place after the "
                               f"forward that is being differentiated (version {p._version}, expected {v}); "
                               "run backward before the optimizer step / in-place update")
    return ctx.params


def _lin_bwd(dz, a, W, pro_a, need_da=True, x3=False):
    """Linear backward pieces for z = pro(a)·W^T + b given dz:
    dW = dz^T·pro(a) (TN, prologue re-applied to the stored pre-activation),
    d pro(a) = dz·W (NN)."""
    dW = ops.gemm(dz, a, TN, pro=pro_a, x3=x3)
    da = ops.gemm(dz, W, NN, x3=x3) if need_da else None
    return dW, da


_side_streams = {}


class _Leaves:
    """Runs the leaves of a backward — gradients no later kernel of the step reads — on a second stream.
    `run(fn, *reads)` makes the side stream wait for everything issued so far on the current one, then calls fn
    under it; `reads` are the tensors fn consumes (their memory must not be recycled by the current stream's
    allocator pool before the side stream is done with them).  `join(*outs)` orders the current stream after the
    side stream and hands the outputs over.  With config.SIDE_STREAM off, run() just calls fn."""

    def __init__(self, device, enable=True):
        self.on = enable and config.SIDE_STREAM and device.type == "cuda"
        if self.on:
            self.main = torch.cuda.current_stream(device)
            self.side = _side_streams.get(device.index)
            if self.side is None:
                self.side = _side_streams[device.index] = torch.cuda.Stream(device)
            self.on = self.side != self.main

    def run(self, fn, *reads):
        if not self.on:
            return fn()
        self.side.wait_stream(self.main)
        for t in reads:
            if t is not None:
                t.record_stream(self.side)
        with torch.cuda.stream(self.side):
            return fn()

    def join(self, *outs):
        if not self.on:
            return
        self.main.wait_stream(self.side)
        for t in outs:
            if t is not None:
                t.record_stream(self.main)


# ===========================================================================
# Encoder
# ===========================================================================
def _split_ok(rows, k, split):
    """A Linear runs as a bf16x3 split GEMM when the mode asks for it, the layer is big
    enough to be MFMA-bound and its reduction width fits the 8-column sx8 groups."""
    return split and rows >= config.SPLIT_MIN_ROWS and k % 8 == 0 and k >= 32


def _tn_either(a_s, b_s):
    """The wgrad kernel tiles dW[Mo, No] with Mo % 256 == 0; when only the transposed problem fits
    (e.g. dW[128, 256]) it computes dW^T = h^T·dz and the small result is transposed."""
    return ops.gemm_split_tn_ok(a_s, b_s) or ops.gemm_split_tn_ok(b_s, a_s)


def _wgrad_tn(dz_s, h_s):
    if ops.gemm_split_tn_ok(dz_s, h_s):
        return ops.gemm_split_tn(dz_s, h_s)
    return ops.gemm_split_tn(h_s, dz_s).t().contiguous()


class EncoderFn(torch.autograd.Function):
    """x[B,N,Din], params -> (pooled[B,2C] = [masked max | masked mean], point_features[B,N,C],
    upooled[B,2C] = [unmasked mean | unmasked max]).

    params = [W_i, b_i, gamma_i, beta_i]*n_hidden + [W_out, b_out]   (the per-point shared MLP)
    upooled holds the UNMASKED pools the vertex head needs, already in its concatenation order
    (VertexPredictor.py:86-88), produced by the same pass as the masked ones.  The fusion MLP is its own Function (FusionFn) so that
    its gradients are handed to autograd — and to the data-parallel reducer — before the long
    per-point backward starts.

    precision "bf16x3": layers whose input width is a multiple of 8 consume the previous
    layer's relu(LN(z)) as an sx8 split operand produced by wf3d_ln_prep (which also
    yields the LN statistics), through wf3d_gemm_split; backward uses split GEMMs for
    dgrad (W^T in sx8) and wgrad (transposed sx8 operands).  Only z and (mu, rstd) are
    kept for backward in either mode."""

    @staticmethod
    def forward(ctx, x, n_hidden, precision, *params):
        # a cloud that requires grad (train.py never asks) sends the first layer's backward down the general path, which
        # forms dz and from it dx = dz . W
        ctx.want_dx = ctx.needs_input_grad[0]
        B, N, Din = x.shape
        M = B * N
        x2 = x.reshape(M, Din)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        split = precision == "bf16x3"
        valid = ops.point_valid(x2)
        # both sx8 orientations (W for the forward GEMM, W^T for its dgrad) of every split layer's weight, one launch
        widx = [i for i in range(1, n_hidden + 1)
                if _split_ok(M, params[4 * i].shape[1], split) and _split_ok(M, params[4 * i].shape[0], split)]
        wsplit = dict(zip(widx, ops.split_weights([params[4 * i] for i in widx]))) if widx else {}
        zs, stats, hs = [], [], []                 # hs[i] = sx8 operand consumed by Linear i+1 (kept for its wgrad)
        a, a_s, pro = x2, None, None
        for i in range(n_hidden + 1):
            last = i == n_hidden
            W, b = params[4 * i], params[4 * i + 1]
            fused0 = (i == 0 and not last and _split_ok(M, W.shape[0], split) and ops.first_layer_ok(x2, W))
            if fused0:
                # first Linear + LayerNorm + ReLU + split in one pass (K = input_dim <= 8: a pure write of z and h)
                z, mu, rs, a_s = ops.first_layer_fwd(x2, W, b, params[2], params[3], ACT_RELU)
            elif a_s is not None:
                z = ops.gemm_split(a_s, wsplit[i][0] if i in wsplit else ops.split_rows(W), bias=b)
                a_s = None
            else:
                z = ops.gemm(a, W, NT, bias=b, pro=pro)
                a_s = None
            if last:
                pf = z
                break
            g, be = params[4 * i + 2], params[4 * i + 3]
            if fused0:
                pass
            elif _split_ok(M, z.shape[1], split):
                mu, rs, a_s = ops.ln_prep(z, g, be, ACT_RELU)
            else:
                mu, rs = ops.row_stats(z)
            hs.append(a_s)
            zs.append(z)
            stats.append((mu, rs))
            pro = Pro(ACT_RELU, mu, rs, g, be)
            a = z
        del a_s
        C = pf.shape[1]
        po = ops.pool4_fwd(pf.view(B, N, C), valid, packed=True)          # [max | avg] (PointNetEncoder.py:115), [mean | max]
        ctx.n_hidden, ctx.dims, ctx.split = n_hidden, (B, N, C), split
        _keep_params(ctx, params)
        ctx.saved = (x2, valid, zs, stats, hs, po.arg_m, po.arg_u, po.cnt)
        ctx.nvalid = po.nvalid
        ctx.wT = {i: ws[1] for i, ws in wsplit.items()}
        # unused outputs (point_features when only its pools are consumed) must not come back as 268 MB of zeros
        ctx.set_materialize_grads(False)
        pf3 = pf.view(B, N, C)
        return po.masked, pf3, po.unmasked

    @staticmethod
    def backward(ctx, dpooled, dpf, dupooled):
        nh, (B, N, C), split = ctx.n_hidden, ctx.dims, ctx.split
        M = B * N
        params = _params(ctx)
        x2, valid, zs, stats, hs, arg_m, arg_u, cnt = _saved(ctx)
        grads = [None] * len(params)
        dx = None
        # the pooled cotangents are read as strided halves of the [B, 2C] gradients: no copies
        dmmax = dmavg = dumean = dumax = None
        if dpooled is not None:
            dpooled = dpooled if dpooled.stride(1) == 1 else dpooled.contiguous()
            dmmax, dmavg = dpooled[:, :C], dpooled[:, C:]
        if dupooled is not None:
            dupooled = dupooled if dupooled.stride(1) == 1 else dupooled.contiguous()
            dumean, dumax = dupooled[:, :C], dupooled[:, C:]
        if dpf is not None:
            dpf = dpf.contiguous()
        # output layer: with both of its consumers on the split path dz goes straight out as an sx8 operand
        # and its column sum (the bias gradient) follows from the [B, C] cotangents
        W_out = params[4 * nh]
        out_split = (nh > 0 and _split_ok(M, W_out.shape[0], split) and C % 8 == 0 and hs[nh - 1] is not None
                     and ops.gemm_split_tn_shape_ok(M, C, hs[nh - 1].shape[1], C, hs[nh - 1].stride(0)))
        if out_split:
            dz, dz_s = None, ops.pool4_bwd(valid, cnt, arg_m, arg_u, dmmax, dmavg, dumean, dumax, dpf, B, N, C, sx8=True).view(M, C)
            grads[4 * nh + 1] = ops.pool4_bwd_colsum(ctx.nvalid, cnt, arg_m, arg_u, dmmax, dmavg, dumean, dumax, dpf, B, N, C)
        else:
            dz = ops.pool4_bwd(valid, cnt, arg_m, arg_u, dmmax, dmavg, dumean, dumax, dpf, B, N, C).view(M, C)
            dz_s = None
        # per-point MLP backward, output layer first.  Entering iteration i: dz = grad wrt the
        # pre-LN output of Linear i (fp32) and, in split mode, dz_s = the same in sx8.
        for i in range(nh, -1, -1):
            W = params[4 * i]
            if i == 0 and nh > 0 and not ctx.want_dx and _split_ok(M, W.shape[0], split) and ops.first_layer_ok(x2, W):
                # layer 0: LN/ReLU backward, bias and WEIGHT gradient in one pass over (dh, z); dz is never written
                g, be = params[2], params[3]
                grads[2], grads[3], grads[1], grads[0] = ops.ln_act_bwd_first(dh, zs[0], x2, stats[0][0], stats[0][1], g, be, ACT_RELU)
                break
            if i < nh:
                g, be = params[4 * i + 2], params[4 * i + 3]
                mu, rs = stats[i]
                want_s = i > 0 and _split_ok(M, W.shape[0], split)
                dz_s = torch.empty_like(dh) if want_s else None
                # with both consumers (dgrad, wgrad) on the split path the fp32 dz is never needed
                tn_ok = want_s and hs[i - 1] is not None and ops.gemm_split_tn_ok(dh, hs[i - 1])
                all_split = want_s and (tn_ok or (_split_ok(M, W.shape[1], split) and M % 8 == 0))
                dz, grads[4 * i + 2], grads[4 * i + 3], grads[4 * i + 1] = ops.ln_act_bwd(
                    dh, zs[i], mu, rs, g, be, ACT_RELU, inplace=True, dz_split=dz_s, want_dz=not all_split)
            elif dz is not None:
                grads[4 * i + 1] = ops.colsum(dz)
                if nh and _split_ok(M, W.shape[0], split):
                    dz_s = ops.split_rows(dz)
            if i > 0:
                pg, pb = params[4 * (i - 1) + 2], params[4 * (i - 1) + 3]
                a_prev, pro_prev = zs[i - 1], Pro(ACT_RELU, stats[i - 1][0], stats[i - 1][1], pg, pb)
            else:
                a_prev, pro_prev = x2, None
            K = W.shape[1]
            h_prev_s = hs[i - 1] if i > 0 else None
            if i > 0 and dz_s is not None and h_prev_s is not None and ops.gemm_split_tn_ok(dz_s, h_prev_s):
                # wgrad dW = dz^T · h_prev straight from the reduction-major sx8 operands (transposing LDS reads)
                grads[4 * i] = ops.gemm_split_tn(dz_s, h_prev_s)
                if not config.RETAIN_SAVED:
                    hs[i - 1] = None
            elif i > 0 and dz_s is not None and _split_ok(M, K, split) and M % 8 == 0:
                # same through materialised transposes (shapes the TN kernel does not tile)
                grads[4 * i] = ops.gemm_split(ops.split_transpose(dz_s, in_sx8=True), ops.split_transpose(a_prev, pro_prev))
            else:
                grads[4 * i] = ops.gemm(dz, a_prev, TN, pro=pro_prev)
            if i > 0:
                if dz_s is not None:
                    dh = ops.gemm_split(dz_s, ctx.wT[i] if i in ctx.wT else ops.split_rows(W, transpose=True))     # dgrad: dz · W
                else:
                    dh = ops.gemm(dz, W, NN)
            elif ctx.want_dx:
                dx = ops.gemm(dz, W, NN).view(B, N, -1)           # at layer 0 dz is always the fp32 tensor
            del dz, dz_s
            dz = dz_s = None
        if not config.RETAIN_SAVED:
            ctx.saved = None
        return (dx, None, None, *grads)


class FusionFn(torch.autograd.Function):
    """pooled[B,2C] -> global[B,C]: feature_fusion (PointNetEncoder.py:57-65,116).
    params = [F0w, F0b, F1g, F1b, F3w, F3b, F4g, F4b, F6w, F6b]

    B <= 32 rows: three launches of the weight-streaming skinny kernel forward, six backward
    (csrc/skinny.hip); otherwise the generic GEMM path."""

    @staticmethod
    def forward(ctx, pooled, *F):
        pooled = pooled if pooled.stride(1) == 1 else pooled.contiguous()
        M = pooled.shape[0]
        _keep_params(ctx, F)
        ctx.skinny = sk.ok(M, F[0], F[4], F[8]) and F[0].shape[0] % 16 == 0 and F[4].shape[0] % 16 == 0 \
            and pooled.stride(0) % 4 == 0
        if ctx.skinny:
            (f0, p0, _), = sk.fwd(M, sk.Fwd(pooled, F[0], F[1], stats=True))
            (f3, p3, s0), = sk.fwd(M, sk.Fwd(f0, F[4], F[5], ln=sk.LNIn(F[2], F[3], ACT_RELU, part=p0), stats=True))
            (gl, _, s3), = sk.fwd(M, sk.Fwd(f3, F[8], F[9], ln=sk.LNIn(F[6], F[7], ACT_RELU, part=p3)))
        else:
            f0 = ops.gemm(pooled, F[0], NT, bias=F[1])
            s0 = ops.row_stats(f0)
            f3 = ops.gemm(f0, F[4], NT, bias=F[5], pro=Pro(ACT_RELU, s0[0], s0[1], F[2], F[3]))
            s3 = ops.row_stats(f3)
            gl = ops.gemm(f3, F[8], NT, bias=F[9], pro=Pro(ACT_RELU, s3[0], s3[1], F[6], F[7]))
        ctx.saved = (pooled, f0, s0, f3, s3)
        return gl

    @staticmethod
    def backward(ctx, dgl):
        F = _params(ctx)
        pooled, f0, s0, f3, s3 = _saved(ctx)
        G = [None] * 10
        dgl = dgl.contiguous()
        if ctx.skinny:
            M = dgl.shape[0]
            l3 = sk.LNIn(F[6], F[7], ACT_RELU, mu=s3[0], rs=s3[1])
            l0 = sk.LNIn(F[2], F[3], ACT_RELU, mu=s0[0], rs=s0[1])
            (G[8], G[9], sl), = sk.bwd(M, sk.Bwd(dgl, F[8], f3, x_ln=l3))
            _, g3, G[6], G[7], rp3 = sk.reduce(M, F[8].shape[1], [sl], want_dh=False, ln=(f3, s3[0], s3[1], F[6], F[7], ACT_RELU))
            (G[4], G[5], sl), = sk.bwd(M, sk.Bwd(g3, F[4], f0, ln_out=sk.LNOut(f3, s3[0], s3[1], rp3), x_ln=l0))
            _, g0, G[2], G[3], rp0 = sk.reduce(M, F[4].shape[1], [sl], want_dh=False, ln=(f0, s0[0], s0[1], F[2], F[3], ACT_RELU))
            (G[0], G[1], sl), = sk.bwd(M, sk.Bwd(g0, F[0], pooled, ln_out=sk.LNOut(f0, s0[0], s0[1], rp0)))
            dpooled, *_ = sk.reduce(M, F[0].shape[1], [sl])
        else:
            G[9] = ops.colsum(dgl)
            G[8], dh = _lin_bwd(dgl, f3, F[8], Pro(ACT_RELU, s3[0], s3[1], F[6], F[7]))
            dz, G[6], G[7], G[5] = ops.ln_act_bwd(dh, f3, s3[0], s3[1], F[6], F[7], ACT_RELU, inplace=True)
            G[4], dh = _lin_bwd(dz, f0, F[4], Pro(ACT_RELU, s0[0], s0[1], F[2], F[3]))
            dz, G[2], G[3], G[1] = ops.ln_act_bwd(dh, f0, s0[0], s0[1], F[2], F[3], ACT_RELU, inplace=True)
            G[0], dpooled = _lin_bwd(dz, pooled, F[0], None)
        if not config.RETAIN_SAVED:
            ctx.saved = None
        return (dpooled, *G)


# ===========================================================================
# Vertex head
# ===========================================================================
class VertexFn(torch.autograd.Function):
    """(g[B,C], upooled[B,2C] = [mean | max] or None, params) -> (vertices[B,V,3] (view of o[B,V,vd]), exist[B,V], counts[B] int64)

    params = [W1,b1,g1,be1, W2,b2,g2,be2, W3,b3,g3,be3, W4,b4,g4,be4, Wf,bf, Wr1,br1, Wr2,br2, (Wpp,bpp)]

    B <= 32 rows: six weight-streaming launches forward (the three Linears that read `e` share one; the residual sums
    c = relu(LN(z3)) + r1, d = relu(LN(z4)) + r2 are applied on load and never materialised), thirteen backward
    (csrc/skinny.hip); otherwise the generic GEMM path."""

    @staticmethod
    def forward(ctx, g, upooled, V, vd, *params):
        g = g.contiguous()
        B = g.shape[0]
        (W1, b1, g1, be1, W2, b2, g2, be2, W3, b3, g3, be3, W4, b4, g4, be4, Wf, bf, Wr1, br1, Wr2, br2) = params[:22]
        if upooled is not None:
            upooled = upooled if upooled.stride(1) == 1 else upooled.contiguous()
        mats = [W1, W2, W3, W4, Wf, Wr1, Wr2] + ([params[22]] if upooled is not None else [])
        ctx.skinny = sk.ok(B, *mats) and all(w.shape[0] % 16 == 0 for w in (W1, W2, W3, W4)) \
            and (upooled is None or upooled.stride(0) % 4 == 0)
        if ctx.skinny:
            if upooled is not None:
                (e, _, _), = sk.fwd(B, sk.Fwd(upooled, params[22], params[23], out_add=g))     # enhanced = g + proj (:99-100)
            else:
                e = g
            (z1, p1, _), (r1, _, _), (r2, _, _) = sk.fwd(B, sk.Fwd(e, W1, b1, stats=True), sk.Fwd(e, Wr1, br1), sk.Fwd(e, Wr2, br2))
            (z2, p2, s1), = sk.fwd(B, sk.Fwd(z1, W2, b2, ln=sk.LNIn(g1, be1, ACT_RELU, part=p1), stats=True))
            (z3, p3, s2), = sk.fwd(B, sk.Fwd(z2, W3, b3, ln=sk.LNIn(g2, be2, ACT_RELU, part=p2), stats=True))
            # residual after the ReLU (:110,:114), applied on load
            (z4, p4, s3), = sk.fwd(B, sk.Fwd(z3, W4, b4, ln=sk.LNIn(g3, be3, ACT_RELU, part=p3), in_add=r1, stats=True))
            (o, _, s4), = sk.fwd(B, sk.Fwd(z4, Wf, bf, ln=sk.LNIn(g4, be4, ACT_RELU, part=p4), in_add=r2))
            c, d = r1, r2
        else:
            if upooled is not None:
                e = ops.gemm(upooled, params[22], NT, bias=params[23], addend=g)
            else:
                e = g
            z1 = ops.gemm(e, W1, NT, bias=b1); s1 = ops.row_stats(z1)
            z2 = ops.gemm(z1, W2, NT, bias=b2, pro=Pro(ACT_RELU, s1[0], s1[1], g1, be1)); s2 = ops.row_stats(z2)
            z3 = ops.gemm(z2, W3, NT, bias=b3, pro=Pro(ACT_RELU, s2[0], s2[1], g2, be2)); s3 = ops.row_stats(z3)
            r1 = ops.gemm(e, Wr1, NT, bias=br1)
            c = ops.ln_act_apply(z3, s3[0], s3[1], g3, be3, ACT_RELU, addend=r1)   # residual after ReLU (:110)
            z4 = ops.gemm(c, W4, NT, bias=b4); s4 = ops.row_stats(z4)
            r2 = ops.gemm(e, Wr2, NT, bias=br2)
            d = ops.ln_act_apply(z4, s4[0], s4[1], g4, be4, ACT_RELU, addend=r2)
            o = ops.gemm(d, Wf, NT, bias=bf)
        exist, counts = ops.vertex_finalize_fwd(o, V, vd)
        _keep_params(ctx, params)
        ctx.dims = (B, V, vd)
        # skinny path: c / d hold r1 / r2 (the sums are rebuilt on load)
        ctx.saved = (upooled, e, z1, s1, z2, s2, z3, s3, c, z4, s4, d)
        ctx.save_for_backward(exist)
        ctx.mark_non_differentiable(counts)
        # `vertices` = the first three channels, a non-contiguous view as in the reference (VertexPredictor.py:122); handing
        # the view out directly spares autograd's slice backward (a zero-fill and a strided copy per step)
        return o.view(B, V, vd)[:, :, :3], exist, counts

    @staticmethod
    def backward(ctx, do3, dexist, _dcounts):
        params = _params(ctx)
        B, V, vd = ctx.dims
        (W1, b1, g1, be1, W2, b2, g2, be2, W3, b3, g3, be3, W4, b4, g4, be4, Wf, bf, Wr1, br1, Wr2, br2) = params[:22]
        pooled, e, z1, s1, z2, s2, z3, s3, c, z4, s4, d = _saved(ctx)
        (exist,) = ctx.saved_tensors
        G = [None] * len(params)
        do = ops.vertex_finalize_bwd(exist, dexist.contiguous() if dexist is not None else None,
                                     do3.contiguous() if do3 is not None else None, B, V, vd)
        if ctx.skinny:
            r1, r2 = c, d
            R = ACT_RELU
            ln = [None, sk.LNIn(g1, be1, R, mu=s1[0], rs=s1[1]), sk.LNIn(g2, be2, R, mu=s2[0], rs=s2[1]),
                  sk.LNIn(g3, be3, R, mu=s3[0], rs=s3[1]), sk.LNIn(g4, be4, R, mu=s4[0], rs=s4[1])]
            # final layer: input d = relu(LN(z4)) + r2
            (G[16], G[17], sl), = sk.bwd(B, sk.Bwd(do, Wf, z4, x_ln=ln[4], x_add=r2))
            dd, G4, G[14], G[15], rp4 = sk.reduce(B, Wf.shape[1], [sl], ln=(z4, s4[0], s4[1], g4, be4, R))
            # residual_proj2 (plain dz = dd) and vertex_mlp4 (input c = relu(LN(z3)) + r1) in one launch
            (G[20], G[21], sl_e2), (G[12], G[13], sl) = sk.bwd(
                B, sk.Bwd(dd, Wr2, e), sk.Bwd(G4, W4, z3, ln_out=sk.LNOut(z4, s4[0], s4[1], rp4), x_ln=ln[3], x_add=r1))
            dc, G3, G[10], G[11], rp3 = sk.reduce(B, W4.shape[1], [sl], ln=(z3, s3[0], s3[1], g3, be3, R))
            (G[18], G[19], sl_e1), (G[8], G[9], sl) = sk.bwd(
                B, sk.Bwd(dc, Wr1, e), sk.Bwd(G3, W3, z2, ln_out=sk.LNOut(z3, s3[0], s3[1], rp3), x_ln=ln[2]))
            _, G2, G[6], G[7], rp2 = sk.reduce(B, W3.shape[1], [sl], want_dh=False, ln=(z2, s2[0], s2[1], g2, be2, R))
            (G[4], G[5], sl), = sk.bwd(B, sk.Bwd(G2, W2, z1, ln_out=sk.LNOut(z2, s2[0], s2[1], rp2), x_ln=ln[1]))
            _, G1, G[2], G[3], rp1 = sk.reduce(B, W2.shape[1], [sl], want_dh=False, ln=(z1, s1[0], s1[1], g1, be1, R))
            (G[0], G[1], sl_e0), = sk.bwd(B, sk.Bwd(G1, W1, e, ln_out=sk.LNOut(z1, s1[0], s1[1], rp1)))
            de, *_ = sk.reduce(B, W1.shape[1], [sl_e0, sl_e1, sl_e2])
            dupooled = None
            if pooled is not None:
                Wpp = params[22]
                (G[22], G[23], sl), = sk.bwd(B, sk.Bwd(de, Wpp, pooled))
                dupooled, *_ = sk.reduce(B, Wpp.shape[1], [sl])
            if not config.RETAIN_SAVED:
                ctx.saved = None
            return (de, dupooled, None, None, *G)
        G[17] = ops.colsum(do)
        G[16], dd = _lin_bwd(do, d, Wf, None)
        # d = relu(LN(z4)) + r2
        G[21] = ops.colsum(dd)
        G[20] = ops.gemm(dd, e, TN)
        de = ops.gemm(dd, Wr2, NN)
        dz4, G[14], G[15], G[13] = ops.ln_act_bwd(dd, z4, s4[0], s4[1], g4, be4, ACT_RELU)
        G[12], dc = _lin_bwd(dz4, c, W4, None)
        # c = relu(LN(z3)) + r1
        G[19] = ops.colsum(dc)
        G[18] = ops.gemm(dc, e, TN)
        ops.gemm(dc, Wr1, NN, out=de, accumulate=True)
        dz3, G[10], G[11], G[9] = ops.ln_act_bwd(dc, z3, s3[0], s3[1], g3, be3, ACT_RELU)
        G[8], dh = _lin_bwd(dz3, z2, W3, Pro(ACT_RELU, s2[0], s2[1], g2, be2))
        dz2, G[6], G[7], G[5] = ops.ln_act_bwd(dh, z2, s2[0], s2[1], g2, be2, ACT_RELU, inplace=True)
        G[4], dh = _lin_bwd(dz2, z1, W2, Pro(ACT_RELU, s1[0], s1[1], g1, be1))
        dz1, G[2], G[3], G[1] = ops.ln_act_bwd(dh, z1, s1[0], s1[1], g1, be1, ACT_RELU, inplace=True)
        G[0] = ops.gemm(dz1, e, TN)
        ops.gemm(dz1, W1, NN, out=de, accumulate=True)
        dupooled = None
        if pooled is not None:
            Wpp = params[22]
            G[23] = ops.colsum(de)
            G[22] = ops.gemm(de, pooled, TN)
            dupooled = ops.gemm(de, Wpp, NN)
        if not config.RETAIN_SAVED:
            ctx.saved = None
        return (de, dupooled, None, None, *G)


# ===========================================================================
# Edge head (all samples of the batch in one pass, ragged)
# ===========================================================================
class EdgeFn(torch.autograd.Function):
    """verts[B,V,3] (view ok) -> edge_probs[B, max_E] zero-padded.

    params = [P0w,P0b, P1g,P1b, P3w,P3b, P4g,P4b, Aw,Ab, Ow,Ob,
              M0w,M0b, M1g,M1b, M4w,M4b, M5g,M5b, M8w,M8b, M10w,M10b]
    (vertex_proj.{0,1,3,4}, attention.{in_proj,out_proj}, edge_mlp.{0,1,4,5,8,10})"""

    @staticmethod
    def forward(ctx, verts, counts, heads, drop_ps, seed, precision, *params):
        (P0w, P0b, P1g, P1b, P3w, P3b, P4g, P4b, Aw, Ab, Ow, Ob,
         M0w, M0b, M1g, M1b, M4w, M4b, M5g, M5b, M8w, M8b, M10w, M10b) = params
        B, V, _ = verts.shape
        H = P3w.shape[0]
        meta = ops.EdgeMeta.get(counts, verts.device)
        sd = [(seed + 0x9E3779B9 * k) & 0xFFFFFFFF for k in range(1, 5)]
        pf_, pa_, p1_, p2_ = (float(x) for x in drop_ps)   # vertex_proj.5, attention, edge_mlp.3, edge_mlp.7
        cv = ops.edge_gather_verts(verts, meta)
        za = ops.gemm(cv, P0w, NT, bias=P0b); sa = ops.row_stats(za)
        # per-vertex Linears (sum-of-counts rows): in bf16x3 mode the fp32 operands are split inside the GEMM's
        # staging pass (x3), forward and backward; the K = 3 coordinate products stay exact fp32
        x3 = precision == "bf16x3" and config.EDGE_X3
        zb = ops.gemm(za, P3w, NT, bias=P3b, pro=Pro(ACT_GELU, sa[0], sa[1], P1g, P1b), x3=x3); sb = ops.row_stats(zb)
        f = ops.ln_act_apply(zb, sb[0], sb[1], P4g, P4b, ACT_NONE, drop_p=pf_, seed=sd[0])
        qkv = ops.gemm(f, Aw, NT, bias=Ab, x3=x3)
        cx, lse = ops.attn_fwd(qkv, meta, H, heads, pa_, sd[1])
        Fm = ops.gemm(cx, Ow, NT, bias=Ob, addend=f, x3=x3)              # residual (EdgePredictor.py:114)
        vd = cv.shape[1]                                   # coordinates per vertex (EdgePredictor(vertex_dim=...): 3 in the model)
        Wa, Wb, Wc, Wd = M0w[:, :H], M0w[:, H:2 * H], M0w[:, 2 * H:2 * H + vd], M0w[:, 2 * H + vd:2 * H + 2 * vd]
        if vd <= 4:
            # the coordinate columns ride on the two GEMMs as a rank-vd epilogue term (exact fp32)
            Pa = ops.gemm(Fm, Wa, NT, bias=M0b, x3=x3, lowrank=(cv, Wc))
            Pb = ops.gemm(Fm, Wb, NT, x3=x3, lowrank=(cv, Wd))
        else:
            Pa = ops.gemm(Fm, Wa, NT, bias=M0b, x3=x3)
            ops.gemm(cv, Wc, NT, out=Pa, accumulate=True)
            Pb = ops.gemm(Fm, Wb, NT, x3=x3)
            ops.gemm(cv, Wd, NT, out=Pb, accumulate=True)
        # the two wide edge-MLP layers (E rows: 52 % of the FLOPs at V=256) on the split path
        # (H / 4, the third layer's width, is the reduction width of its dgrad and the row length of its transposed weight:
        # it has to be made of whole 8-column groups too — EdgePredictor(hidden_dim=208) used to fail in split_weights)
        split = _split_ok(meta.Re, H, precision == "bf16x3") and _split_ok(meta.Re, H // 2, True) and (H // 4) % 8 == 0
        if split:
            # the pair kernel holds each edge row in registers: it also emits gelu(LN(pre)) as the next GEMM's operand.
            # `pre` itself (2 KB per edge row) is stored only when something in backward reads it: with the wgrad of the
            # second edge layer on the transposing-read kernel (operand h1) and the LayerNorm backward rebuilding the
            # row from Pa / Pb (ops.edge_pair_ln_bwd), nothing does.
            tn2 = (ops.gemm_split_tn_shape_ok(meta.Re, H // 2, H, H // 2, H)
                   or ops.gemm_split_tn_shape_ok(meta.Re, H, H // 2, H, H // 2))          # what backward's _tn_either(dh2, h1) will say
            keep_pre = not (tn2 and H <= 1024) or config.KEEP_PRE
            pre, mu0, rs0, delta, h1 = ops.edge_pair_fwd(Pa, Pb, cv, M0w, meta, ln=(M1g, M1b, ACT_GELU, p1_, sd[2]), keep_pre=keep_pre)
        else:
            pre, mu0, rs0, delta = ops.edge_pair_fwd(Pa, Pb, cv, M0w, meta)
        if pre is not None:
            Pa = Pb = None
        if split:
            (M4s, M4t), (M8s, M8t) = ops.split_weights([M4w, M8w])
            ctx.wT = (M4t, M8t)
            z2 = ops.gemm_split(h1, M4s, bias=M4b)
            mu2, rs2, h2 = ops.ln_prep(z2, M5g, M5b, ACT_GELU, drop_p=p2_, seed=sd[3])
            s2 = (mu2, rs2)
            z3 = ops.gemm_split(h2, M8s, bias=M8b)
        else:
            h1 = h2 = None
            z2 = ops.gemm(pre, M4w, NT, bias=M4b, pro=Pro(ACT_GELU, mu0, rs0, M1g, M1b, p1_, sd[2]), x3=x3); s2 = ops.row_stats(z2)
            z3 = ops.gemm(z2, M8w, NT, bias=M8b, pro=Pro(ACT_GELU, s2[0], s2[1], M5g, M5b, p2_, sd[3]), x3=x3)
        if ops.rowdot_act_ok(z3, M10w):
            logit = ops.rowdot_act(z3, M10w, M10b, ACT_GELU)          # one-output Linear: a row dot product, not a GEMM
        else:
            logit = ops.gemm(z3, M10w, NT, bias=M10b, pro=Pro(ACT_GELU))
        probs = ops.edge_prob_fwd(logit, meta)
        _keep_params(ctx, params)
        ctx.cfg = (B, V, H, heads, (pf_, pa_, p1_, p2_), sd, meta)
        ctx.split, ctx.x3 = split, x3
        ctx.saved = (cv, za, sa, zb, sb, f, qkv, cx, lse, Fm, pre, mu0, rs0, delta, z2, s2, z3, h1, h2, Pa, Pb)
        ctx.save_for_backward(probs)
        return probs

    @staticmethod
    def backward(ctx, dprobs):
        params = _params(ctx)
        (P0w, P0b, P1g, P1b, P3w, P3b, P4g, P4b, Aw, Ab, Ow, Ob,
         M0w, M0b, M1g, M1b, M4w, M4b, M5g, M5b, M8w, M8b, M10w, M10b) = params
        B, V, H, heads, (pf_, pa_, p1_, p2_), sd, meta = ctx.cfg
        cv, za, sa, zb, sb, f, qkv, cx, lse, Fm, pre, mu0, rs0, delta, z2, s2, z3, h1, h2, Pa, Pb = _saved(ctx)
        (probs,) = ctx.saved_tensors
        G = [None] * len(params)
        x3 = ctx.x3
        # weight / bias gradients run beside the dgrad chain while the launches are small enough to share the chip
        # (cfg2, 64.5 k edge rows: +0.7 % on the step; cfg5, 1.04 M rows: -1.3 %, every kernel fills it alone)
        lv = _Leaves(dprobs.device, enable=meta.Re <= (1 << 18))
        dlogit = ops.edge_prob_bwd(probs, dprobs.contiguous(), meta)                      # [Re,1]
        G[23] = ops.colsum(dlogit)
        fused_tail = ops.rowdot_act_ok(z3, M10w)
        if fused_tail:
            dh3 = z3            # shape / dtype stand-in: the fused kernel below produces dz3 directly
        elif M10w.shape[0] == 1:
            # one-output Linear: dW[c] = sum_r dlogit[r]*gelu(z3[r,c]) is a weighted column sum,
            # d gelu(z3) = dlogit (outer) W — no 128x128 MFMA tile wasted on a 1-wide problem
            G[22] = ops.colsum(z3, dlogit.view(-1), ACT_GELU).view(1, -1)
            dh3 = ops.gemm(dlogit, M10w, NN)
        else:
            G[22], dh3 = _lin_bwd(dlogit, z3, M10w, Pro(ACT_GELU))
        p2 = Pro(ACT_GELU, s2[0], s2[1], M5g, M5b, p2_, sd[3])
        p1 = Pro(ACT_GELU, mu0, rs0, M1g, M1b, p1_, sd[2])
        tsplit = ctx.split and meta.Re % 8 == 0              # (transposed) wgrad operands need whole 8-row groups
        if ctx.split:
            dz3_s = torch.empty_like(dh3)
            tn3 = _tn_either(dh3, h2)                       # dz3_s has dh3's shape
            if fused_tail:
                # logits-layer backward + GELU backward + both column sums in one pass over z3
                dz3, G[22], G[21] = ops.rowdot_act_bwd(z3, dlogit, M10w, ACT_GELU, want_dz=not (tsplit or tn3), dz_split=dz3_s)
            else:
                dz3, _, _, G[21] = ops.ln_act_bwd(dh3, z3, None, None, None, None, ACT_GELU, inplace=True,
                                                  dz_split=dz3_s, want_dz=not (tsplit or tn3))
            if tn3:
                G[20] = lv.run(lambda: _wgrad_tn(dz3_s, h2), dz3_s, h2)
            else:
                G[20] = lv.run(lambda: (ops.gemm_split(ops.split_transpose(dz3_s, in_sx8=True), ops.split_transpose(z2, p2)) if tsplit
                                        else ops.gemm(dz3, z2, TN, pro=p2, x3=x3)), dz3_s, dz3, z2)
            dh2 = ops.gemm_split(dz3_s, ctx.wT[1])
            del dz3, dz3_s
            dz2_s = torch.empty_like(dh2)
            tn2 = _tn_either(dh2, h1)
            dz2, G[18], G[19], G[17] = ops.ln_act_bwd(dh2, z2, s2[0], s2[1], M5g, M5b, ACT_GELU, p2_, sd[3],
                                                      inplace=True, dz_split=dz2_s, want_dz=not (tsplit or tn2))
            if tn2:
                G[16] = lv.run(lambda: _wgrad_tn(dz2_s, h1), dz2_s, h1)
            else:
                G[16] = lv.run(lambda: (ops.gemm_split(ops.split_transpose(dz2_s, in_sx8=True), ops.split_transpose(pre, p1)) if tsplit
                                        else ops.gemm(dz2, pre, TN, pro=p1, x3=x3)), dz2_s, dz2, pre)
            dh1 = ops.gemm_split(dz2_s, ctx.wT[0])
            del dz2, dz2_s
        else:
            if fused_tail:
                dz3, G[22], G[21] = ops.rowdot_act_bwd(z3, dlogit, M10w, ACT_GELU)
            else:
                dz3, _, _, G[21] = ops.ln_act_bwd(dh3, z3, None, None, None, None, ACT_GELU, inplace=True)
            G[20] = lv.run(lambda: ops.gemm(dz3, z2, TN, pro=p2, x3=x3), dz3, z2)
            dh2 = ops.gemm(dz3, M8w, NN, x3=x3)
            dz2, G[18], G[19], G[17] = ops.ln_act_bwd(dh2, z2, s2[0], s2[1], M5g, M5b, ACT_GELU, p2_, sd[3], inplace=True)
            G[16] = lv.run(lambda: ops.gemm(dz2, pre, TN, pro=p1, x3=x3), dz2, pre)
            dh1 = ops.gemm(dz2, M4w, NN, x3=x3)
        # LN/GELU backward of the first edge layer; the same pass yields the gradient of its distance-weight column
        if pre is None:
            dpre, G[14], G[15], wsum = ops.edge_pair_ln_bwd(dh1, Pa, Pb, delta, M0w, meta, mu0, rs0, M1g, M1b, ACT_GELU, p1_, sd[2])
        else:
            dpre, G[14], G[15], wsum = ops.ln_act_bwd_wsum(dh1, pre, delta, mu0, rs0, M1g, M1b, ACT_GELU, p1_, sd[2], inplace=True)
        # split first layer backward
        vd = cv.shape[1]
        dW0 = torch.empty_like(M0w)           # every column is written below: Wa | Wb | Wc | Wd | w_delta
        dW0[:, 2 * H + 2 * vd].copy_(wsum)
        dPa, dPb, dcv = ops.edge_pair_bwd(dpre, delta, cv, M0w, meta, coord=True)     # dcv includes dPa·Wc + dPb·Wd

        def first_layer_leaves():
            ops.gemm(dPa, Fm, TN, out=dW0[:, :H], x3=x3)
            ops.gemm(dPb, Fm, TN, out=dW0[:, H:2 * H], x3=x3)
            ops.gemm(dPa, cv, TN, out=dW0[:, 2 * H:2 * H + vd])
            ops.gemm(dPb, cv, TN, out=dW0[:, 2 * H + vd:2 * H + 2 * vd])
            return ops.colsum(dPa)
        G[13] = lv.run(first_layer_leaves, dPa, dPb, Fm, cv, dW0)
        G[12] = dW0
        Wa, Wb = M0w[:, :H], M0w[:, H:2 * H]
        dF = ops.gemm(dPa, Wa, NN, x3=x3)
        ops.gemm(dPb, Wb, NN, out=dF, accumulate=True, x3=x3)
        # F = f + out_proj(ctx)
        G[11], G[10] = lv.run(lambda: (ops.colsum(dF), ops.gemm(dF, cx, TN, x3=x3)), dF, cx)
        dcx = ops.gemm(dF, Ow, NN, x3=x3)
        dqkv = ops.attn_bwd(qkv, dcx, cx, lse, meta, H, heads, pa_, sd[1])
        G[9], G[8] = lv.run(lambda: (ops.colsum(dqkv), ops.gemm(dqkv, f, TN, x3=x3)), dqkv, f)
        df = ops.gemm(dqkv, Aw, NN, addend=dF, x3=x3)
        # f = drop(LN(zb))
        dzb, G[6], G[7], G[5] = ops.ln_act_bwd(df, zb, sb[0], sb[1], P4g, P4b, ACT_NONE, pf_, sd[0], inplace=True)
        pa = Pro(ACT_GELU, sa[0], sa[1], P1g, P1b)
        G[4] = lv.run(lambda: ops.gemm(dzb, za, TN, pro=pa, x3=x3), dzb, za)
        dha = ops.gemm(dzb, P3w, NN, x3=x3)
        dza, G[2], G[3], G[1] = ops.ln_act_bwd(dha, za, sa[0], sa[1], P1g, P1b, ACT_GELU, inplace=True)
        G[0] = lv.run(lambda: ops.gemm(dza, cv, TN), dza, cv)
        ops.gemm(dza, P0w, NN, out=dcv, accumulate=True)
        lv.join(G[20], G[16], G[13], dW0, G[11], G[10], G[9], G[8], G[4], G[0])
        dverts = ops.edge_scatter_dverts(dcv, meta, B, V)
        if not config.RETAIN_SAVED:
            ctx.saved = None
        return (dverts, None, None, None, None, None, *G)


class _FrozenList(list):
    """A list that refuses in-place mutation.  edge_index_lists hands the SAME cached object to every
    sample and every forward with that vertex count (the reference builds fresh lists each call,
    EdgePredictor.py:88-89,140 — 63 ms per sample at V=256); a caller that filtered one in place would
    corrupt every later forward, so mutation raises instead.  Compares equal to / converts like a list."""
    __slots__ = ()

    def _ro(self, *a, **k):
        raise TypeError("edge_indices lists are shared between calls and read-only: copy first "
                        "(e.g. [list(p) for p in out['edge_indices'][i]])")

    append = extend = insert = remove = pop = clear = sort = reverse = _ro
    __setitem__ = __delitem__ = __iadd__ = __imul__ = _ro


_EDGE_LIST_CACHE = {}          # vertex count -> _FrozenList of _FrozenList([i, j]); LRU, bounded
_EDGE_LIST_CACHE_MAX = 128     # 2.6 MB of host memory per entry at V=256


def edge_index_lists(counts):
    """Per-sample list of [i, j] pairs, i < j, lexicographic — bit-identical to
    EdgePredictor._get_edge_indices(...).tolist() (reference :70-89,140), but
    built once per vertex count and cached (LRU) instead of re-looped on every call."""
    out = []
    for c in counts:
        lst = _EDGE_LIST_CACHE.pop(c, None)
        if lst is None:
            lst = _FrozenList(_FrozenList((i, j)) for i in range(c) for j in range(i + 1, c))
            if len(_EDGE_LIST_CACHE) >= _EDGE_LIST_CACHE_MAX:
                _EDGE_LIST_CACHE.pop(next(iter(_EDGE_LIST_CACHE)))
        _EDGE_LIST_CACHE[c] = lst          # (re-)insert as most recently used
        out.append(lst)
    return out


class UnmaskedPoolFn(torch.autograd.Function):
    """point_features[B,N,C] -> [mean_n | max_n] as one [B, 2C] tensor: the vertex head's own pooling and its
    concatenation (VertexPredictor.py:86-88) when it is used outside PointCloudToWireframe."""

    @staticmethod
    def forward(ctx, pf):
        pf = pf.contiguous()
        B, N, C = pf.shape
        ones = torch.ones(B * N, dtype=torch.float32, device=pf.device)
        po = ops.pool4_fwd(pf, ones, packed=True)
        ctx.dims, ctx.saved = (B, N, C), (ones, po.cnt, po.arg_m, po.arg_u)
        return po.unmasked

    @staticmethod
    def backward(ctx, dup):
        B, N, C = ctx.dims
        ones, cnt, arg_m, arg_u = _saved(ctx)
        dup = dup if dup.stride(1) == 1 else dup.contiguous()
        return ops.pool4_bwd(ones, cnt, arg_m, arg_u, None, None, dup[:, :C], dup[:, C:], None, B, N, C)
