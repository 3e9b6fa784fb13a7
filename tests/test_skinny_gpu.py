"""csrc/skinny.hip (Linear layers over M <= 32 rows: feature-fusion MLP and vertex head, reference
models/PointNetEncoder.py:57-65 and models/VertexPredictor.py:94-117) against fp64 torch math of the same ops,
through the C ABI.  Forward 1e-5 of the tensor scale (exact-fp32 MFMA chains), gradients element-wise 1e-4 of
max(|g|, rms(g)); the ReLU decisions are the kernels' own (no LayerNorm output within 1e-6 of 0 in these inputs)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import helpers as H  # noqa: E402

RELU, NONE = 1, 0


def dev():
    return torch.device("cuda:0")


def rnd(gen, *shape, scale=1.0):
    return (torch.randn(*shape, generator=gen) * scale).to(dev())


def ln64(x, g, b, eps=1e-5):
    mu = x.mean(1, keepdim=True)
    var = ((x - mu) ** 2).mean(1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * g + b


@pytest.mark.parametrize("M", [1, 8, 17, 32])
@pytest.mark.parametrize("K,N1,N2", [(64, 160, 104), (1000, 2048, 256), (16, 32, 20)])
def test_forward_chain_with_merged_statistics(M, K, N1, N2):
    """Y1 = X·W1^T + b1 (+ stats partials);  Y2 = (relu(LN(Y1)) + A)·W2^T + b2 + R, and a second Linear on X in
    the same launch as the first."""
    from wf3d import skinny as sk
    gen = torch.Generator().manual_seed(M * 1000 + K)
    X, W1, b1 = rnd(gen, M, K), rnd(gen, N1, K, scale=K ** -0.5), rnd(gen, N1)
    Wx, bx = rnd(gen, 48, K, scale=K ** -0.5), rnd(gen, 48)
    g, be = 1 + 0.1 * rnd(gen, N1), 0.1 * rnd(gen, N1)
    A, W2, b2, R = rnd(gen, M, N1), rnd(gen, N2, N1, scale=N1 ** -0.5), rnd(gen, N2), rnd(gen, M, N2)
    assert sk.ok(M, W1, W2, Wx)
    (Y1, part, _), (Yx, _, _) = sk.fwd(M, sk.Fwd(X, W1, b1, stats=True), sk.Fwd(X, Wx, bx))
    (Y2, _, (mu, rs)), = sk.fwd(M, sk.Fwd(Y1, W2, b2, ln=sk.LNIn(g, be, RELU, part=part), in_add=A, out_add=R))
    X6, W6 = X.double(), W1.double()
    r1 = X6 @ W6.t() + b1.double()
    assert H.rel_err(Y1.cpu().numpy(), r1.cpu().numpy()) < 1e-5
    assert H.rel_err(Yx.cpu().numpy(), (X6 @ Wx.double().t() + bx.double()).cpu().numpy()) < 1e-5
    y1 = Y1.double()
    assert H.rel_err(mu.cpu().numpy(), y1.mean(1).cpu().numpy()) < 1e-5
    assert H.rel_err(rs.cpu().numpy(), (1 / torch.sqrt(y1.var(1, unbiased=False) + 1e-5)).cpu().numpy()) < 1e-5
    r2 = (torch.relu(ln64(y1, g.double(), be.double())) + A.double()) @ W2.double().t() + b2.double() + R.double()
    assert H.rel_err(Y2.cpu().numpy(), r2.cpu().numpy()) < 1e-5


@pytest.mark.parametrize("M", [2, 32])
@pytest.mark.parametrize("K,N1,N2", [(64, 160, 104), (512, 2048, 1024), (20, 32, 12)])
def test_backward_chain(M, K, N1, N2):
    """Two Linears with LayerNorm+ReLU and a residual between them, plus a side Linear that shares the input:
         z1 = X·W1^T + b1 ; h = relu(LN(z1)) + r ; r = X·Wr^T + br ; y = h·W2^T + b2 ; loss = sum(y * C)
    against fp64 autograd: every weight / bias / LayerNorm gradient and dX."""
    from wf3d import skinny as sk
    gen = torch.Generator().manual_seed(M * 77 + K)
    X = rnd(gen, M, K)
    W1, b1 = rnd(gen, N1, K, scale=K ** -0.5), rnd(gen, N1)
    Wr, br = rnd(gen, N1, K, scale=K ** -0.5), rnd(gen, N1)
    g, be = 1 + 0.1 * rnd(gen, N1), 0.1 * rnd(gen, N1)
    W2, b2 = rnd(gen, N2, N1, scale=N1 ** -0.5), rnd(gen, N2)
    C = rnd(gen, M, N2)
    # forward
    (z1, part, _), (r, _, _) = sk.fwd(M, sk.Fwd(X, W1, b1, stats=True), sk.Fwd(X, Wr, br))
    (y, _, (mu, rs)), = sk.fwd(M, sk.Fwd(z1, W2, b2, ln=sk.LNIn(g, be, RELU, part=part), in_add=r))
    # backward
    xl = sk.LNIn(g, be, RELU, mu=mu, rs=rs)
    (dW2, db2, sl_h), = sk.bwd(M, sk.Bwd(C, W2, z1, x_ln=xl, x_add=r))
    dh, G, dg, dbe, rowpart = sk.reduce(M, N1, [sl_h], ln=(z1, mu, rs, g, be, RELU))
    (dW1, db1, sl_x1), (dWr, dbr, sl_xr) = sk.bwd(
        M, sk.Bwd(G, W1, X, ln_out=sk.LNOut(z1, mu, rs, rowpart)), sk.Bwd(dh, Wr, X))
    dX, *_ = sk.reduce(M, K, [sl_x1, sl_xr])
    # fp64 reference
    P = {k: v.double().cpu().requires_grad_() for k, v in dict(X=X, W1=W1, b1=b1, Wr=Wr, br=br, g=g, be=be, W2=W2, b2=b2).items()}
    z1r = P["X"] @ P["W1"].t() + P["b1"]
    v = ln64(z1r, P["g"], P["be"])
    assert int((v.abs() < 1e-6).sum()) == 0            # no borderline ReLU decision in this input
    rr = P["X"] @ P["Wr"].t() + P["br"]
    yr = (torch.relu(v) + rr) @ P["W2"].t() + P["b2"]
    (yr * C.double().cpu()).sum().backward()
    assert H.rel_err(y.cpu().numpy(), yr.detach().numpy()) < 1e-5
    got = dict(X=dX, W1=dW1, b1=db1, Wr=dWr, br=dbr, g=dg, be=dbe, W2=dW2, b2=db2)
    for k, t in got.items():
        e = H.elem_err(t.cpu().numpy(), P[k].grad.numpy())
        assert e < 1e-4, (k, e)


def test_reduce_is_deterministic_and_sums_three_sets():
    from wf3d import skinny as sk
    gen = torch.Generator().manual_seed(3)
    M, K = 5, 200
    sets = [rnd(gen, n, M, K) for n in (3, 1, 7)]
    extra = rnd(gen, M, K)
    a, *_ = sk.reduce(M, K, sets, extra=extra)
    b, *_ = sk.reduce(M, K, sets, extra=extra)
    assert torch.equal(a, b)
    want = sum(s.double().sum(0) for s in sets) + extra.double()
    assert H.rel_err(a.cpu().numpy(), want.cpu().numpy()) < 1e-6


def test_unsupported_shapes_are_refused():
    from wf3d import skinny as sk
    W = torch.zeros(8, 6, device=dev())
    assert not sk.ok(4, W)                              # K % 4 != 0
    assert not sk.ok(33, torch.zeros(8, 8, device=dev()))
    with pytest.raises(RuntimeError, match="skinny_fwd"):
        sk.fwd(4, sk.Fwd(torch.zeros(4, 6, device=dev()), W, None))
