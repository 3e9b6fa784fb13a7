"""Per-kernel parity of the HIP ops (through the C ABI) against plain PyTorch
references of the same op.  fp32 tolerance: 1e-4 relative to the tensor's
scale for anything that went through a GEMM (BASELINE.json north_star), 2e-5
for elementwise / reduction kernels."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

import helpers as H  # noqa: E402,F401  (sets sys.path)

TOL_GEMM = 1e-4
TOL_ELT = 2e-5


@pytest.fixture(scope="module")
def ops():
    from wf3d import ops as o
    o._lib.load()
    return o


def dev():
    return torch.device("cuda:0")


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    if a.numel() == 0:
        return 0.0
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed + 1000 * len(shape) + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).to(dev())


def ln_ref(z, gamma, beta, act):
    y = torch.nn.functional.layer_norm(z, (z.shape[-1],), gamma, beta, 1e-5)
    if act == 1:
        y = torch.relu(y)
    elif act == 2:
        y = torch.nn.functional.gelu(y)
    return y


def ref64(fn, *ts):
    return fn(*[t.double().cpu() if torch.is_tensor(t) else t for t in ts])


@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (300, 200, 100), (32, 130, 64), (5, 7, 3), (257, 1, 128),
                                   (1000, 512, 8), (64, 96, 1031), (129, 257, 33)])
@pytest.mark.parametrize("layout", [0, 1, 2])
def test_gemm_layouts(ops, M, N, K, layout):
    A = rnd(M, K, seed=1)
    B = rnd(N, K, seed=2)
    bias = rnd(N, seed=3)
    want = ref64(lambda a, b, c: a @ b.T + c, A, B, bias)
    if layout == ops.NT:
        got = ops.gemm(A, B, ops.NT, bias=bias)
    elif layout == ops.NN:
        got = ops.gemm(A, B.T.contiguous(), ops.NN, bias=bias)
    else:
        got = ops.gemm(A.T.contiguous(), B.T.contiguous(), ops.TN, bias=bias)
    assert got.shape == (M, N)
    assert rel(got, want) < TOL_GEMM


def test_gemm_identity_asymmetric(ops):
    # A = I with an asymmetric B catches a transposed C/D fragment map
    n = 160
    eye = torch.eye(n, device=dev())
    B = (torch.arange(n * n, device=dev(), dtype=torch.float32).reshape(n, n) % 97) / 7.0
    got = ops.gemm(eye, B, ops.NT)          # I · B^T
    assert torch.equal(got, B.T.contiguous())
    got = ops.gemm(eye, B, ops.NN)
    assert torch.equal(got, B)
    got = ops.gemm(eye, B, ops.TN)
    assert torch.equal(got, B)


def test_gemm_strided_views_scalar_path(ops):
    # weight column slices of a [512, 1031] matrix (ld = 1031, unaligned): edge split layer
    W = rnd(96, 1031, seed=5)
    X = rnd(70, 512, seed=6)
    Wa = W[:, 512:1024]
    got = ops.gemm(X, Wa, ops.NT)
    assert rel(got, ref64(lambda x, w: x @ w.T, X, Wa)) < TOL_GEMM
    G = rnd(70, 96, seed=7)
    got = ops.gemm(G, Wa, ops.NN)            # dX = G · Wa
    assert rel(got, ref64(lambda g, w: g @ w, G, Wa)) < TOL_GEMM
    dW = torch.zeros_like(W)
    ops.gemm(G, X, ops.TN, out=dW[:, 512:1024])   # dWa = G^T · X into a strided view
    assert rel(dW[:, 512:1024], ref64(lambda g, x: g.T @ x, G, X)) < TOL_GEMM
    assert float(dW[:, :512].abs().max()) == 0.0 and float(dW[:, 1024:].abs().max()) == 0.0


@pytest.mark.parametrize("M,N,K", [(32, 256, 4096), (8, 128, 2048), (128, 128, 8192)])
def test_gemm_splitk_epilogue(ops, M, N, K):
    A, B, bias, R = rnd(M, K, seed=1), rnd(N, K, seed=2), rnd(N, seed=3), rnd(M, N, seed=4)
    C0 = rnd(M, N, seed=5)
    assert ops._lib.load().wf3d_gemm_ws_bytes(M, N, K, 0) > 0       # split-K path is taken
    out = C0.clone()
    ops.gemm(A, B, ops.NT, bias=bias, addend=R, out=out, accumulate=True)
    want = ref64(lambda a, b, c, r, c0: a @ b.T + c + r + c0, A, B, bias, R, C0)
    assert rel(out, want) < TOL_GEMM
    # TN with a long reduction (wgrad shape)
    G, X = rnd(K, 96, seed=6), rnd(K, 200, seed=7)
    got = ops.gemm(G, X, ops.TN)
    assert rel(got, ref64(lambda g, x: g.T @ x, G, X)) < TOL_GEMM


@pytest.mark.parametrize("act", [0, 1, 2])
@pytest.mark.parametrize("M,N,K", [(200, 96, 256), (33, 130, 68), (512, 512, 512)])
def test_gemm_ln_prologue_nt_and_tn(ops, act, M, N, K):
    Z = rnd(M, K, seed=1, scale=2.0) + 0.3
    gamma, beta = 1 + 0.2 * rnd(K, seed=2), 0.1 * rnd(K, seed=3)
    W = rnd(N, K, seed=4)
    mu, rs = ops.row_stats(Z)
    m64 = Z.double().mean(1)
    assert rel(mu, m64) < TOL_ELT
    assert rel(rs, 1 / torch.sqrt(Z.double().var(1, unbiased=False) + 1e-5)) < TOL_ELT
    Hh = ref64(lambda z, g, b: ln_ref(z, g, b, act), Z, gamma, beta)
    pro = ops.Pro(act, mu, rs, gamma, beta)
    got = ops.gemm(Z, W, ops.NT, pro=pro)
    assert rel(got, Hh @ W.double().cpu().T) < TOL_GEMM
    # wgrad form: dW[N_out, K] = G^T · act(LN(Z)) — prologue on the B operand
    G = rnd(M, N, seed=5)
    got = ops.gemm(G, Z, ops.TN, pro=pro)
    assert rel(got, G.double().cpu().T @ Hh) < TOL_GEMM
    # materialising form agrees too
    Hm = ops.ln_act_apply(Z, mu, rs, gamma, beta, act)
    assert rel(Hm, Hh) < TOL_ELT


def test_gemm_prologue_no_ln_gelu(ops):
    Z, W = rnd(100, 128, seed=1), rnd(1, 128, seed=2)
    got = ops.gemm(Z, W, ops.NT, pro=ops.Pro(ops.ACT_GELU))
    want = ref64(lambda z, w: torch.nn.functional.gelu(z) @ w.T, Z, W)
    assert rel(got, want) < TOL_GEMM


def test_dropout_mask_consistency(ops):
    """Prologue dropout, materialised dropout and the backward kernel must use the same mask."""
    M, K, N, p, seed = 150, 256, 64, 0.1, 12345
    Z = rnd(M, K, seed=1)
    gamma, beta = 1 + 0.2 * rnd(K, seed=2), 0.1 * rnd(K, seed=3)
    W = rnd(N, K, seed=4)
    mu, rs = ops.row_stats(Z)
    Hd = ops.ln_act_apply(Z, mu, rs, gamma, beta, ops.ACT_GELU, drop_p=p, seed=seed)
    Hn = ops.ln_act_apply(Z, mu, rs, gamma, beta, ops.ACT_GELU)
    keep = (Hd != 0) | (Hn == 0)
    frac = 1 - keep.float().mean().item()
    assert abs(frac - p) < 0.02
    assert rel(Hd[keep], Hn[keep] / (1 - p)) < TOL_ELT
    got = ops.gemm(Z, W, ops.NT, pro=ops.Pro(ops.ACT_GELU, mu, rs, gamma, beta, p, seed))
    assert rel(got, Hd.double().cpu() @ W.double().cpu().T) < TOL_GEMM
    got = ops.gemm(rnd(M, N, seed=9), Z, ops.TN, pro=ops.Pro(ops.ACT_GELU, mu, rs, gamma, beta, p, seed))
    assert rel(got, rnd(M, N, seed=9).double().cpu().T @ Hd.double().cpu()) < TOL_GEMM
    # backward: autograd through the same mask
    dh = rnd(M, K, seed=5)
    Zc = Z.double().cpu().requires_grad_()
    g64, b64 = gamma.double().cpu().requires_grad_(), beta.double().cpu().requires_grad_()
    mask = (keep.double().cpu() / (1 - p))
    y = ln_ref(Zc, g64, b64, 2) * mask
    y.backward(dh.double().cpu())
    dz, dg, db, dbias = ops.ln_act_bwd(dh, Z, mu, rs, gamma, beta, ops.ACT_GELU, drop_p=p, seed=seed)
    assert rel(dz, Zc.grad) < 5 * TOL_ELT
    assert rel(dg, g64.grad) < 5 * TOL_ELT and rel(db, b64.grad) < 5 * TOL_ELT
    assert rel(dbias, Zc.grad.sum(0)) < 1e-4


@pytest.mark.parametrize("act", [0, 1, 2])
@pytest.mark.parametrize("R,D", [(37, 32), (300, 512), (64, 2048), (5, 4096), (1025, 1024), (3, 16), (9, 264), (11, 516), (7, 1032), (5, 4088)])
def test_ln_act_bwd(ops, act, R, D):
    Z = rnd(R, D, seed=1, scale=1.5) + 0.2
    gamma, beta = 1 + 0.2 * rnd(D, seed=2), 0.1 * rnd(D, seed=3)
    dh = rnd(R, D, seed=4)
    mu, rs = ops.row_stats(Z)
    Zc = Z.double().cpu().requires_grad_()
    g64, b64 = gamma.double().cpu().requires_grad_(), beta.double().cpu().requires_grad_()
    ln_ref(Zc, g64, b64, act).backward(dh.double().cpu())
    dz, dg, db, dbias = ops.ln_act_bwd(dh, Z, mu, rs, gamma, beta, act)
    assert rel(dz, Zc.grad) < 5 * TOL_ELT
    assert rel(dg, g64.grad) < 5 * TOL_ELT
    assert rel(db, b64.grad) < 5 * TOL_ELT
    assert rel(dbias, Zc.grad.sum(0)) < 2e-4          # LN backward rows sum to ~0: compare on dz scale
    # in place, and the activation-only form (no LayerNorm: edge_mlp.8 -> GELU)
    d2 = dh.clone()
    dz2, _, _, _ = ops.ln_act_bwd(d2, Z, mu, rs, gamma, beta, act, inplace=True)
    assert dz2.data_ptr() == d2.data_ptr() and torch.equal(dz2, dz)
    Zc2 = Z.double().cpu().requires_grad_()
    a = Zc2 if act == 0 else (torch.relu(Zc2) if act == 1 else torch.nn.functional.gelu(Zc2))
    a.backward(dh.double().cpu())
    dz3, dg3, db3, dbias3 = ops.ln_act_bwd(dh, Z, None, None, None, None, act)
    assert dg3 is None and db3 is None
    assert rel(dz3, Zc2.grad) < TOL_ELT and rel(dbias3, Zc2.grad.sum(0)) < 5 * TOL_ELT


@pytest.mark.parametrize("R,D", [(1, 5), (63, 100), (5000, 512), (131, 2048)])
def test_colsum(ops, R, D):
    X, w = rnd(R, D, seed=1), rnd(R, seed=2)
    assert rel(ops.colsum(X), X.double().sum(0)) < TOL_ELT
    assert rel(ops.colsum(X, w), (X.double() * w.double()[:, None]).sum(0)) < TOL_ELT
    want = (torch.nn.functional.gelu(X.double()) * w.double()[:, None]).sum(0)
    assert rel(ops.colsum(X, w, ops.ACT_GELU), want) < TOL_ELT


@pytest.mark.parametrize("B,N,C", [(3, 37, 16), (2, 1000, 512), (1, 4096, 512), (5, 33, 300), (2, 19, 8), (2, 50, 24), (1, 23, 1032), (2, 12, 2048), (2, 31, 768), (3, 13, 40)])
def test_pool4_fwd_bwd(ops, B, N, C):
    x = rnd(B, N, 8, seed=1)
    x[:, ::7] = 0.0                       # zero-padded points
    if B > 1:
        x[1] = 0.0                        # one fully padded cloud
    pf = rnd(B, N, C, seed=2)
    pf[0, 3] = pf[0, 11]                  # exact ties: first index must win
    valid = ops.point_valid(x)
    vm = x.abs().sum(-1) > 1e-9
    assert torch.equal(valid.reshape(B, N) > 0, vm)
    po = ops.pool4_fwd(pf, valid, packed=True)
    mmax, mavg, umean, umax, arg_m, arg_u, cnt = po.mmax, po.mavg, po.umean, po.umax, po.arg_m, po.arg_u, po.cnt
    # the packed [B, 2C] vectors are the reference's torch.cat orders (PointNetEncoder.py:115, VertexPredictor.py:88)
    assert torch.equal(po.masked, torch.cat([mmax, mavg], 1)) and torch.equal(po.unmasked, torch.cat([umean, umax], 1))
    assert torch.equal(po.nvalid.cpu(), vm.sum(1).float().cpu())
    # reference formulation (oracle.encoder_pools / VertexPredictor pools)
    from helpers import oracle
    pfc = pf.cpu().requires_grad_()
    mx_r, avg_r = oracle.encoder_pools(x.cpu(), pfc)
    um_r, ux = pfc.mean(1), pfc.max(1)
    assert torch.equal(mmax.cpu(), mx_r.detach())                       # max is exact
    assert torch.equal(umax.cpu(), ux.values.detach())
    assert rel(mavg, avg_r.detach()) < TOL_ELT and rel(umean, um_r.detach()) < TOL_ELT
    assert torch.equal(arg_u.cpu().long(), ux.indices)
    assert torch.equal(cnt.cpu(), vm.sum(1).clamp(min=1).float().cpu())
    cot = [rnd(B, C, seed=10 + i) for i in range(4)]
    direct = rnd(B, N, C, seed=20)
    (mx_r * cot[0].cpu() + avg_r * cot[1].cpu() + um_r * cot[2].cpu() + ux.values * cot[3].cpu()).sum().backward()
    dpf = ops.pool4_bwd(valid, cnt, arg_m, arg_u, cot[0], cot[1], cot[2], cot[3], None, B, N, C)
    assert rel(dpf, pfc.grad) < TOL_ELT
    dpf2 = ops.pool4_bwd(valid, cnt, arg_m, arg_u, cot[0], cot[1], cot[2], cot[3], direct, B, N, C)
    assert rel(dpf2, pfc.grad + direct.cpu()) < TOL_ELT
    # column sums from the [B, C] cotangents alone == column sums of the scattered tensor
    for d, full in ((None, dpf), (direct, dpf2)):
        cs = ops.pool4_bwd_colsum(po.nvalid, cnt, arg_m, arg_u, cot[0], cot[1], cot[2], cot[3], d, B, N, C)
        assert rel(cs, full.double().sum((0, 1))) < 2e-5
    # cotangents as the two halves of [B, 2C] gradients (what the fusion MLP / vertex head hand back)
    gm, gu = torch.cat([cot[0], cot[1]], 1), torch.cat([cot[2], cot[3]], 1)
    dpf3 = ops.pool4_bwd(valid, cnt, arg_m, arg_u, gm[:, :C], gm[:, C:], gu[:, :C], gu[:, C:], None, B, N, C)
    assert torch.equal(dpf3, dpf)
    cs3 = ops.pool4_bwd_colsum(po.nvalid, cnt, arg_m, arg_u, gm[:, :C], gm[:, C:], gu[:, :C], gu[:, C:], None, B, N, C)
    assert rel(cs3, dpf.double().sum((0, 1))) < 2e-5
    if C % 8 == 0:
        # sx8 output: hi + lo planes reproduce the fp32 result to the split format's 2^-17
        for d, full in ((None, dpf), (direct, dpf2)):
            s8 = ops.pool4_bwd(valid, cnt, arg_m, arg_u, cot[0], cot[1], cot[2], cot[3], d, B, N, C, sx8=True)
            sh, sl = unpack_sx8(s8.view(B * N, C))
            assert torch.equal(sh, full.view(B * N, C).bfloat16().float())
            assert rel(sh + sl, full.view(B * N, C)) < 2e-5


def test_cpu_tensors_rejected(ops):
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.gemm(torch.zeros(4, 4), torch.zeros(4, 4), ops.NT)


def test_bad_arguments_raise(ops):
    with pytest.raises(RuntimeError):
        ops.gemm(rnd(4, 5), rnd(4, 6), ops.NT)
    with pytest.raises(RuntimeError):
        ops.ln_act_bwd(rnd(4, 6), rnd(4, 6), None, None, None, None, 0)    # D % 4 != 0


# ---------------------------------------------------------------------------
# bf16x3 split-precision path (sx8 operands)
# ---------------------------------------------------------------------------
def unpack_sx8(t):
    """sx8 tensor [R, C] (fp32 container) -> (hi, lo) float32 [R, C]."""
    R, C = t.shape
    b = t.contiguous().view(torch.bfloat16).reshape(R, C // 8, 2, 8)
    return b[:, :, 0, :].reshape(R, C).float(), b[:, :, 1, :].reshape(R, C).float()


def ref_split(x):
    hi = x.to(torch.bfloat16).float()
    lo = (x - hi).to(torch.bfloat16).float()
    return hi, lo


TOL_SPLIT = 3e-5      # bf16x3 keeps ~16 significant bits per product; gate on the path is 1e-4


@pytest.mark.parametrize("R,C", [(5, 8), (300, 512), (64, 1032)])
def test_split_rows_format_and_transpose(ops, R, C):
    X = rnd(R, C, seed=1, scale=3.0)
    hi, lo = unpack_sx8(ops.split_rows(X))
    rh, rl = ref_split(X)
    assert torch.equal(hi, rh) and torch.equal(lo, rl)                  # bit-exact RNE split
    assert rel(hi + lo, X) < 2 ** -15
    if R % 8 == 0:
        th, tl = unpack_sx8(ops.split_rows(X, transpose=True))
        assert torch.equal(th, rh.T) and torch.equal(tl, rl.T)
    V = X[:, : (C // 16) * 8] if C >= 16 else X                          # strided view input
    vh, _ = unpack_sx8(ops.split_rows(V))
    assert torch.equal(vh, ref_split(V.contiguous())[0])


@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (300, 200, 104), (1000, 512, 512), (64, 96, 2048), (4096, 1024, 1024)])
def test_gemm_split_matches_fp64(ops, M, N, K):
    A, B, bias = rnd(M, K, seed=1), rnd(N, K, seed=2), rnd(N, seed=3)
    got = ops.gemm_split(ops.split_rows(A), ops.split_rows(B), bias=bias)
    want = ref64(lambda a, b, c: a @ b.T + c, A, B, bias)
    assert rel(got, want) < TOL_SPLIT
    C0 = rnd(M, N, seed=4)
    out = C0.clone()
    ops.gemm_split(ops.split_rows(A), ops.split_rows(B), out=out, accumulate=True)
    assert rel(out, want - bias.double().cpu() + C0.double().cpu()) < TOL_SPLIT
    # dgrad form: dX = dY · W  ==  gemm_split(dY_s, (W^T)_s)
    G = rnd(M, N, seed=5)
    got = ops.gemm_split(ops.split_rows(G), ops.split_rows(B, transpose=True)) if N % 8 == 0 else None
    if got is not None:
        assert rel(got, ref64(lambda g, w: g @ w, G, B)) < TOL_SPLIT


def test_gemm_split_long_k_splitk(ops):
    M, N, K = 96, 200, 65536           # wgrad-like: few tiles, long reduction -> split-K slabs
    A, B = rnd(M, K, seed=1), rnd(N, K, seed=2)
    assert ops._lib.load().wf3d_gemm_split_ws_bytes(M, N, K) > 0
    got = ops.gemm_split(ops.split_rows(A), ops.split_rows(B))
    assert rel(got, ref64(lambda a, b: a @ b.T, A, B)) < TOL_SPLIT


@pytest.mark.parametrize("act", [0, 1, 2])
@pytest.mark.parametrize("R,D", [(37, 32), (300, 512), (129, 2048), (5, 4096),
                                 # widths that leave slots partly empty / exactly full, both lane groupings, odd row counts
                                 (1, 8), (3, 136), (77, 256), (9, 264), (131, 768), (64, 1024), (5, 1032), (33, 1536)])
def test_ln_prep_and_bwd_split_output(ops, act, R, D):
    Z = rnd(R, D, seed=1, scale=1.5) + 0.2
    gamma, beta = 1 + 0.2 * rnd(D, seed=2), 0.1 * rnd(D, seed=3)
    mu, rs, hs = ops.ln_prep(Z, gamma, beta, act)
    mu2, rs2 = ops.row_stats(Z)
    assert rel(mu, mu2) < 1e-6 and rel(rs, rs2) < 1e-6
    hi, lo = unpack_sx8(hs)
    want = ref64(lambda z, g, b: ln_ref(z, g, b, act), Z, gamma, beta)
    assert rel(hi + lo, want) < 2e-5
    h32 = ops.ln_act_apply(Z, mu, rs, gamma, beta, act)
    rh, rl = ref_split(h32)
    assert (hi - rh).abs().max() <= 2 ** -7 * h32.abs().max()           # same split up to 1-ulp LN differences
    dh = rnd(R, D, seed=4)
    ds = torch.empty_like(Z)
    dz, _, _, _ = ops.ln_act_bwd(dh, Z, mu, rs, gamma, beta, act, dz_split=ds)
    dh_, dl_ = unpack_sx8(ds)
    eh, el = ref_split(dz)
    assert torch.equal(dh_, eh) and torch.equal(dl_, el)                 # sx8 copy of dz is the exact split of dz
    ds2 = torch.zeros_like(Z)
    none, dg2, _, _ = ops.ln_act_bwd(dh, Z, mu, rs, gamma, beta, act, dz_split=ds2, want_dz=False)
    assert none is None and torch.equal(ds2, ds)                         # fp32 dz not written, sx8 identical


@pytest.mark.parametrize("act", [0, 1])
@pytest.mark.parametrize("R,C", [(64, 64), (200, 36), (1000, 512), (72, 1028)])
def test_split_transpose_and_wgrad_form(ops, act, R, C):
    Z = rnd(R, C, seed=1, scale=1.5) + 0.1
    th, tl = unpack_sx8(ops.split_transpose(Z))
    rh, rl = ref_split(Z)
    assert torch.equal(th, rh.T) and torch.equal(tl, rl.T)
    gamma, beta = 1 + 0.2 * rnd(C, seed=2), 0.1 * rnd(C, seed=3)
    mu, rs = ops.row_stats(Z)
    pro = ops.Pro(act, mu, rs, gamma, beta)
    ph, pl = unpack_sx8(ops.split_transpose(Z, pro))
    want = ref64(lambda z, g, b: ln_ref(z, g, b, act), Z, gamma, beta)
    assert rel((ph + pl).T, want) < 2e-5
    if C % 8 == 0:
        sh, sl = unpack_sx8(ops.split_transpose(ops.split_rows(Z), in_sx8=True))     # sx8 -> sx8 transpose
        assert torch.equal(sh + sl, (rh + rl).T)
    # wgrad: dW[N, C] = G^T · act(LN(Z))  via NT-form split GEMM on the transposed operands
    N = 96
    G = rnd(R, N, seed=5)
    dW = ops.gemm_split(ops.split_transpose(G), ops.split_transpose(Z, pro))
    assert rel(dW, G.double().cpu().T @ want) < TOL_SPLIT


@pytest.mark.parametrize("K", [32, 96, 512])
def test_gemm_split_tall_exact_on_integer_data(ops, K):
    """>= 512 tiles of 256x256: the 16x16x32-MFMA kernel at the encoder's launch shape.  Integer data makes every
    fp32 sum exact whatever the order, so any slice counted twice / dropped / mis-addressed shows up bit-exactly."""
    M, N = 32768, 1024
    A = (torch.arange(M * K, device=dev()).reshape(M, K) % 11 - 5).float()
    B = ((torch.arange(N * K, device=dev()).reshape(N, K) * 7) % 13 - 6).float()
    bias = (torch.arange(N, device=dev()) % 5 - 2).float()
    got = ops.gemm_split(ops.split_rows(A), ops.split_rows(B), bias=bias)
    want = A.double() @ B.double().T + bias.double()
    assert torch.equal(got.double(), want)
    # random data, same launch shape: within the split tolerance of fp64
    Ar, Br = rnd(M, K, seed=11), rnd(N, K, seed=12)
    gr = ops.gemm_split(ops.split_rows(Ar), ops.split_rows(Br))
    assert rel(gr[::37], (Ar[::37].double() @ Br.double().T).cpu()) < TOL_SPLIT


@pytest.mark.parametrize("K,Mo,No", [(32, 256, 128), (64, 256, 128), (4096, 512, 256), (1024, 256, 384), (131072, 256, 128),
                                     (32, 256, 256), (96, 512, 256), (4000 * 32 // 32 * 32 // 125, 256, 512), (65536, 512, 256)])
def test_gemm_split_tn_matches_transposed_nt_path(ops, K, Mo, No):
    """wgrad on reduction-major operands (transposing LDS reads) == the NT kernel on materialised
    transposes, and == fp64 within the split tolerance."""
    A, B = rnd(K, Mo, seed=1), rnd(K, No, seed=2)
    As, Bs = ops.split_rows(A), ops.split_rows(B)
    assert ops.gemm_split_tn_ok(As, Bs)
    got = ops.gemm_split_tn(As, Bs)
    want = ref64(lambda a, b: a.T @ b, A, B)
    assert rel(got, want) < TOL_SPLIT
    via_t = ops.gemm_split(ops.split_transpose(As, in_sx8=True), ops.split_transpose(Bs, in_sx8=True))
    assert rel(got, via_t) < 5e-6           # same products, different tile / split-K summation order
    # asymmetric integer data catches any lane / row mix-up of the transposing reads exactly
    Ai = (torch.arange(K * Mo, device=dev()).reshape(K, Mo) % 13 - 6).float()
    Bi = (torch.arange(K * No, device=dev()).reshape(K, No) % 7 - 3).float()
    if K <= 4096:
        gi = ops.gemm_split_tn(ops.split_rows(Ai), ops.split_rows(Bi))
        assert torch.equal(gi.cpu().double(), Ai.double().cpu().T @ Bi.double().cpu())
    assert not ops.gemm_split_tn_ok(ops.split_rows(rnd(64, 136, seed=3)), Bs[:64])


@pytest.mark.parametrize("R,K,D", [(1000, 8, 512), (37, 3, 64), (4096, 8, 1024), (5, 8, 8), (33, 8, 520), (7, 5, 264), (19999, 8, 512)])
def test_first_layer_fused_forward_and_backward(ops, R, K, D):
    """Layer 0 of the per-point MLP: fused Linear+LN+ReLU+split forward == GEMM then ln_prep; fused backward
    (LN/ReLU backward + bias + weight gradient, no dz) == ln_act_bwd + TN GEMM."""
    x = rnd(R, K, seed=1)
    x[::7] = 0.0
    W, b = rnd(D, K, seed=2, scale=0.3), rnd(D, seed=3, scale=0.1)
    gamma, beta = 1.0 + 0.1 * rnd(D, seed=4), 0.1 * rnd(D, seed=5)
    assert ops.first_layer_ok(x, W)
    z, mu, rs, hs = ops.first_layer_fwd(x, W, b, gamma, beta, ops.ACT_RELU)
    z_ref = x.double().cpu() @ W.double().cpu().T + b.double().cpu()
    assert rel(z, z_ref) < TOL_ELT
    mu2, rs2, hs2 = ops.ln_prep(z, gamma, beta, ops.ACT_RELU)
    assert rel(mu, mu2) < 1e-5 and rel(rs, rs2) < 1e-6                    # same z, row sums in a different order
    (h1, l1), (h2, l2) = unpack_sx8(hs), unpack_sx8(hs2)
    assert rel(h1 + l1, h2 + l2) < 2e-5
    dh = rnd(R, D, seed=6)
    dg, db, dbias, dW = ops.ln_act_bwd_first(dh, z, x, mu, rs, gamma, beta, ops.ACT_RELU)
    dz, dg2, db2, dbias2 = ops.ln_act_bwd(dh.clone(), z, mu, rs, gamma, beta, ops.ACT_RELU)
    assert rel(dg, dg2) < 1e-5 and rel(db, db2) < 1e-5 and rel(dbias, dbias2) < 2e-5
    assert rel(dW, dz.double().cpu().T @ x.double().cpu()) < 2e-5


def test_wgrad_through_the_transposed_problem(ops):
    """dW[128, 256] = dz^T·h does not tile (Mo % 256 != 0) but its transpose does: functional._wgrad_tn computes
    h^T·dz and transposes the small result (third edge-MLP layer, EdgePredictor.py:64)."""
    from wf3d import functional as F
    K, Mo, No = 4096, 128, 256
    dz, h = rnd(K, Mo, seed=1), rnd(K, No, seed=2)
    dz_s, h_s = ops.split_rows(dz), ops.split_rows(h)
    assert not ops.gemm_split_tn_ok(dz_s, h_s) and ops.gemm_split_tn_ok(h_s, dz_s) and F._tn_either(dz_s, h_s)
    got = F._wgrad_tn(dz_s, h_s)
    assert got.shape == (Mo, No) and got.is_contiguous()
    assert rel(got, dz.double().cpu().T @ h.double().cpu()) < TOL_SPLIT


@pytest.mark.parametrize("H,counts,drop", [(512, [9, 2, 5], 0.0), (64, [4, 7], 0.0), (512, [6, 6], 0.1), (1024, [3, 8], 0.0)])
def test_pair_forward_also_emits_the_next_operand(ops, H, counts, drop):
    """edge_pair_fwd(ln=...) == edge_pair_fwd followed by ln_prep on the row it wrote (incl. the dropout mask)."""
    meta = ops.EdgeMeta(counts, dev())
    Pa, Pb = rnd(meta.Rv, H, seed=1), rnd(meta.Rv, H, seed=2)
    cv = rnd(meta.Rv, 3, seed=3)
    W0 = rnd(H, 2 * H + 7, seed=4, scale=0.2)
    gamma, beta = 1.0 + 0.1 * rnd(H, seed=5), 0.1 * rnd(H, seed=6)
    pre, mu, rs, delta = ops.edge_pair_fwd(Pa, Pb, cv, W0, meta)
    pre2, mu2, rs2, delta2, h = ops.edge_pair_fwd(Pa, Pb, cv, W0, meta, ln=(gamma, beta, ops.ACT_GELU, drop, 1234))
    assert torch.equal(pre, pre2) and torch.equal(mu, mu2) and torch.equal(rs, rs2) and torch.equal(delta, delta2)
    mu3, rs3, h3 = ops.ln_prep(pre, gamma, beta, ops.ACT_GELU, drop_p=drop, seed=1234)
    (a, b), (c, d) = unpack_sx8(h), unpack_sx8(h3)
    assert rel(a + b, c + d) < 2e-5
    if drop:
        assert torch.equal((a + b) == 0, (c + d) == 0)          # identical dropout mask


@pytest.mark.parametrize("R,D", [(1000, 128), (37, 8), (4099, 512), (64, 64)])
def test_rowdot_logits_layer_forward_and_fused_backward(ops, R, D):
    """One-output Linear on gelu(z): row-dot forward == GEMM with GELU prologue; fused backward == the NN GEMM +
    activation backward + weighted / plain column sums it replaces."""
    z, W, b = rnd(R, D, seed=1), rnd(1, D, seed=2, scale=0.3), rnd(1, seed=3)
    assert ops.rowdot_act_ok(z, W)
    out = ops.rowdot_act(z, W, b, ops.ACT_GELU)
    ref = torch.nn.functional.gelu(z.double().cpu()) @ W.double().cpu().T + b.double().cpu()
    assert out.shape == (R, 1) and rel(out, ref) < 2e-5
    t = rnd(R, 1, seed=4)
    dzs = torch.empty_like(z)
    dz, dW, dbz = ops.rowdot_act_bwd(z, t, W, ops.ACT_GELU, want_dz=True, dz_split=dzs)
    zc = z.double().cpu().requires_grad_()
    (torch.nn.functional.gelu(zc) @ W.double().cpu().T * t.double().cpu()).sum().backward()
    assert rel(dz, zc.grad) < 2e-5
    hi, lo = unpack_sx8(dzs)
    assert rel(hi + lo, zc.grad) < 3e-5
    assert rel(dW, (t.double().cpu().T @ torch.nn.functional.gelu(z.double().cpu()))) < 2e-5
    assert rel(dbz, zc.grad.sum(0)) < 2e-5
    assert not ops.rowdot_act_ok(rnd(8, 24, seed=5), rnd(1, 24, seed=6))       # 24/8 = 3 lanes per row: not a power of two


@pytest.mark.parametrize("H,counts,drop", [(512, [9, 2, 5], 0.0), (64, [4, 7], 0.0), (512, [6, 6], 0.1), (1024, [3, 8], 0.0),
                                           (512, [40, 2, 17, 33], 0.0)])
def test_pair_ln_backward_without_the_stored_preactivation(ops, H, counts, drop):
    """edge_pair_fwd(keep_pre=False) + edge_pair_ln_bwd (pre rebuilt from Pa / Pb in the backward kernel) against the
    stored-pre path (edge_pair_fwd + ln_act_bwd_wsum on it): the same kernel body on the same floats — bit-identical dz,
    column sums to rounding — and the forward side writes the same statistics and the same operand h."""
    meta = ops.EdgeMeta(counts, dev())
    Pa, Pb = rnd(meta.Rv, H, seed=1), rnd(meta.Rv, H, seed=2)
    cv = rnd(meta.Rv, 3, seed=3)
    W0 = rnd(H, 2 * H + 7, seed=4, scale=0.2)
    gamma, beta = 1.0 + 0.1 * rnd(H, seed=5), 0.1 * rnd(H, seed=6)
    ln = (gamma, beta, ops.ACT_GELU, drop, 1234)
    pre, mu, rs, delta, h = ops.edge_pair_fwd(Pa, Pb, cv, W0, meta, ln=ln)
    none, mu2, rs2, delta2, h2 = ops.edge_pair_fwd(Pa, Pb, cv, W0, meta, ln=ln, keep_pre=False)
    assert none is None and torch.equal(mu, mu2) and torch.equal(rs, rs2) and torch.equal(delta, delta2) and torch.equal(h, h2)
    dh = rnd(meta.Re, H, seed=7)
    dz, dg, db, wsum = ops.ln_act_bwd_wsum(dh.clone(), pre, delta, mu, rs, gamma, beta, ops.ACT_GELU, drop, 1234)
    dz2, dg2, db2, wsum2 = ops.edge_pair_ln_bwd(dh.clone(), Pa, Pb, delta, W0, meta, mu, rs, gamma, beta, ops.ACT_GELU, drop, 1234)
    assert torch.equal(dz, dz2)
    assert rel(dg2, dg) < 1e-6 and rel(db2, db) < 1e-6 and rel(wsum2, wsum) < 1e-6


@pytest.mark.parametrize("R,D,drop", [(777, 512, 0.0), (100, 64, 0.1), (300, 1024, 0.0), (64, 2048, 0.0)])
def test_ln_act_bwd_with_weighted_column_sum(ops, R, D, drop):
    """ln_act_bwd_wsum == ln_act_bwd, plus wsum == colsum(dz, wrow) without the second pass."""
    z, dh, wrow = rnd(R, D, seed=1), rnd(R, D, seed=2), rnd(R, seed=3).abs()
    gamma, beta = 1.0 + 0.1 * rnd(D, seed=4), 0.1 * rnd(D, seed=5)
    mu, rs = ops.row_stats(z)
    dz, dg, db, _ = ops.ln_act_bwd(dh.clone(), z, mu, rs, gamma, beta, ops.ACT_GELU, drop, 77, want_bias=False)
    dz2, dg2, db2, wsum = ops.ln_act_bwd_wsum(dh.clone(), z, wrow, mu, rs, gamma, beta, ops.ACT_GELU, drop, 77)
    assert torch.equal(dz, dz2) and rel(dg, dg2) < 1e-6 and rel(db, db2) < 1e-6
    assert rel(wsum, (dz.double().cpu() * wrow.double().cpu()[:, None]).sum(0)) < 2e-5


def test_split_gemms_accumulate_into_out_at_encoder_launch_shapes(ops):
    """accumulate=True on the persistent forward/dgrad kernel (>= 512 full tiles) and on the XCD-mapped split-K wgrad
    kernel: C += A·B^T / A^T·B on integer data, exact."""
    M, N, K = 32768, 1024, 96
    A = (torch.arange(M * K, device=dev()).reshape(M, K) % 7 - 3).float()
    B = ((torch.arange(N * K, device=dev()).reshape(N, K) * 5) % 11 - 5).float()
    out = (torch.arange(M * N, device=dev()).reshape(M, N) % 17 - 8).float()
    want = out.double() + A.double() @ B.double().T
    ops.gemm_split(ops.split_rows(A), ops.split_rows(B), out=out, accumulate=True)
    assert torch.equal(out.double(), want)
    Kt, Mo, No = 65536, 512, 256
    At = (torch.arange(Kt * Mo, device=dev()).reshape(Kt, Mo) % 5 - 2).float()
    Bt = ((torch.arange(Kt * No, device=dev()).reshape(Kt, No) * 3) % 7 - 3).float()
    out2 = (torch.arange(Mo * No, device=dev()).reshape(Mo, No) % 9 - 4).float()
    want2 = out2.double() + At.double().T @ Bt.double()
    ops.gemm_split_tn(ops.split_rows(At), ops.split_rows(Bt), out=out2, accumulate=True)
    assert torch.equal(out2.double(), want2)


# ---- x3: bf16x3 arithmetic on fp32 operands, split while they are staged into LDS ---------------------------------
@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (300, 200, 100), (2048, 512, 512), (2048, 256, 512), (2048, 512, 3),
                                   (129, 257, 33), (1000, 512, 8), (65, 96, 1031), (8192, 1536, 512), (512, 512, 4096)])
@pytest.mark.parametrize("layout", [0, 1, 2])
def test_gemm_x3_layouts(ops, M, N, K, layout):
    A = rnd(M, K, seed=11)
    B = rnd(N, K, seed=12)
    bias = rnd(N, seed=13)
    want = ref64(lambda a, b, c: a @ b.T + c, A, B, bias)
    if layout == ops.NT:
        got = ops.gemm(A, B, ops.NT, bias=bias, x3=True)
    elif layout == ops.NN:
        got = ops.gemm(A, B.T.contiguous(), ops.NN, bias=bias, x3=True)
    else:
        got = ops.gemm(A.T.contiguous(), B.T.contiguous(), ops.TN, bias=bias, x3=True)
    assert rel(got, want) < TOL_SPLIT


@pytest.mark.parametrize("layout", [0, 1, 2])
@pytest.mark.parametrize("M,N,K", [(192, 160, 72), (2048, 192, 40), (77, 130, 19)])
def test_gemm_x3_exact_on_integer_data(ops, layout, M, N, K):
    # small integers are exact in bf16 (lo part 0) and their products/sums exact in fp32: any mis-placed
    # element of the transposing LDS image shows up as an exact mismatch
    g = torch.Generator(device="cpu").manual_seed(5)
    A = torch.randint(-8, 9, (M, K), generator=g).float().to(dev())
    B = torch.randint(-8, 9, (N, K), generator=g).float().to(dev())
    want = (A.double() @ B.double().T).float()
    if layout == ops.NT:
        got = ops.gemm(A, B, ops.NT, x3=True)
    elif layout == ops.NN:
        got = ops.gemm(A, B.T.contiguous(), ops.NN, x3=True)
    else:
        got = ops.gemm(A.T.contiguous(), B.T.contiguous(), ops.TN, x3=True)
    assert torch.equal(got, want)


@pytest.mark.parametrize("act", [0, 2])
@pytest.mark.parametrize("M,N,K", [(2048, 512, 256), (333, 130, 68)])
def test_gemm_x3_ln_prologue_addend_accumulate(ops, act, M, N, K):
    z = rnd(M, K, seed=21, scale=2.0) + 0.5
    W = rnd(N, K, seed=22)
    gamma, beta = rnd(K, seed=23) * 0.3 + 1.0, rnd(K, seed=24) * 0.2
    add = rnd(M, N, seed=25)
    mu, rs = ops.row_stats(z)
    pro = ops.Pro(act, mu, rs, gamma, beta)
    h = ref64(lambda zz, g_, b_: ln_ref(zz, g_, b_, act), z, gamma, beta)
    want = h @ W.double().cpu().T + add.double().cpu()
    got = ops.gemm(z, W, ops.NT, addend=add, pro=pro, x3=True)
    assert rel(got, want) < TOL_SPLIT
    ops.gemm(z, W, ops.NT, out=got, accumulate=True, pro=pro, x3=True)          # got = 2*(h W^T) + add
    assert rel(got, 2 * (h @ W.double().cpu().T) + add.double().cpu()) < TOL_SPLIT
    # TN: dW[N,K] = dz^T · pro(z)
    dz = rnd(M, N, seed=26)
    gw = ops.gemm(dz, z, ops.TN, pro=pro, x3=True)
    assert rel(gw, dz.double().cpu().T @ h) < TOL_SPLIT


@pytest.mark.parametrize("layout", [0, 1, 2])
@pytest.mark.parametrize("M,N,K,k,x3", [(2048, 512, 512, 3, True), (300, 200, 100, 3, False), (33, 130, 64, 4, False),
                                        (128, 128, 8192, 1, False), (2048, 1536, 512, 2, True)])
def test_gemm_lowrank_epilogue(ops, layout, M, N, K, k, x3):
    """C = A.B^T + bias + U.V^T with the rank-k (k <= 4) term added in the epilogue (exact fp32), incl. the split-K
    combine paths and strided U / V views (column slices of wider matrices, as the edge head passes them)."""
    A, B, bias = rnd(M, K, seed=31), rnd(N, K, seed=32), rnd(N, seed=33)
    Uw, Vw = rnd(M, 7, seed=34), rnd(N, 9, seed=35)
    U, V = Uw[:, 2:2 + k], Vw[:, 5:5 + k]
    want = ref64(lambda a, b, c, u, v: a @ b.T + c + u @ v.T, A, B, bias, U.contiguous(), V.contiguous())
    if layout == ops.NT:
        got = ops.gemm(A, B, ops.NT, bias=bias, lowrank=(U, V), x3=x3)
    elif layout == ops.NN:
        got = ops.gemm(A, B.T.contiguous(), ops.NN, bias=bias, lowrank=(U, V), x3=x3)
    else:
        got = ops.gemm(A.T.contiguous(), B.T.contiguous(), ops.TN, bias=bias, lowrank=(U, V), x3=x3)
    assert rel(got, want) < (TOL_SPLIT if x3 else TOL_GEMM)
