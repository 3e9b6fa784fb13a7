"""Stage-level and whole-model parity of the HIP path (through the C ABI and the
drop-in classes) against (a) the golden fixtures produced from the reference and
(b) the CPU oracle on the same inputs.  Tolerance (BASELINE.json north_star):
1e-4 relative fp32 on outputs, bit-exact edge index lists; gradients are held
to 1e-3 of each tensor's scale (they pass through arg-max routing and ~10 chained
GEMMs in a different summation order)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import helpers as H  # noqa: E402
from helpers import detgen, oracle  # noqa: E402

TOL_OUT = 1e-4
TOL_GRAD = 1e-3


def dev():
    return torch.device("cuda:0")


def load_arrays(module, arrs, prefix=""):
    sd = module.state_dict()
    with torch.no_grad():
        for k, v in sd.items():
            v.copy_(torch.from_numpy(arrs[prefix + k]))


@pytest.fixture(params=["fp32", "bf16x3"])
def precision(request):
    """Run a test in both arithmetic modes; bf16x3 is forced onto small shapes too."""
    from wf3d import config
    old = (config.precision(), config.SPLIT_MIN_ROWS)
    config.set_precision(request.param)
    config.SPLIT_MIN_ROWS = 1
    yield request.param
    config.set_precision(old[0])
    config.SPLIT_MIN_ROWS = old[1]


def test_small_encoder_vs_golden(precision):
    from models.PointNetEncoder import PointNetEncoder
    gold = H.load_golden("small_enc")
    shapes = H.sub_shapes("encoder.", input_dim=8, hidden_dims=(32, 64), output_dim=16)
    enc = PointNetEncoder(8, [32, 64], 16).to(dev())
    load_arrays(enc, detgen.fill_state_dict(shapes, 3))
    x = torch.from_numpy(H.make_cloud("small_enc.x", 3, 37, 3, pad_frac=0.2, dead_cloud=2)).to(dev())
    g, pf = enc(x)
    assert H.rel_err(g.detach().cpu().numpy(), gold["out.global"]) < TOL_OUT
    assert H.rel_err(pf.detach().cpu().numpy(), gold["out.point_features"]) < TOL_OUT
    cg = torch.from_numpy(detgen.uniform("small_enc.cot.g", tuple(g.shape), -1, 1, 3)).to(dev())
    cp = torch.from_numpy(detgen.uniform("small_enc.cot.pf", tuple(pf.shape), -1, 1, 3)).to(dev())
    ((g * cg).sum() + (pf * cp).sum()).backward()
    for n, p in enc.named_parameters():
        assert H.rel_err(p.grad.cpu().numpy(), gold["grad." + n]) < TOL_GRAD, n


@pytest.mark.parametrize("V", [7, 2])
def test_small_edge_vs_golden(V, precision):
    from models.EdgePredictor import EdgePredictor
    gold = H.load_golden("small_edge")
    shapes = H.sub_shapes("edge_predictor.", edge_hidden=64)
    ep = EdgePredictor(3, 64, 2).to(dev())
    load_arrays(ep, detgen.fill_state_dict(shapes, 4))
    for sub in ep.modules():
        if isinstance(sub, torch.nn.Dropout):
            sub.p = 0.0
    ep.attention.dropout = 0.0
    ep.train()
    v = torch.from_numpy(detgen.normalish(f"small_edge.v{V}", (2, V, 3), 4)).to(dev()).requires_grad_()
    probs, idx = ep(v)
    assert np.array_equal(np.array(idx, dtype=np.int64), gold[f"V{V}.idx"])
    assert np.array_equal(ep._get_edge_indices(V).cpu().numpy(), gold[f"V{V}.idx"])
    assert H.rel_err(probs.detach().cpu().numpy(), gold[f"V{V}.probs"]) < TOL_OUT
    c = torch.from_numpy(detgen.uniform(f"small_edge.cot{V}", tuple(probs.shape), -1, 1, 4)).to(dev())
    (probs * c).sum().backward()
    assert H.rel_err(v.grad.cpu().numpy(), gold[f"V{V}.dverts"]) < TOL_GRAD
    for n, p in ep.named_parameters():
        key = f"V{V}.grad.{n}"
        if key in gold:
            assert H.rel_err(p.grad.cpu().numpy(), gold[key]) < TOL_GRAD, n
        else:
            assert p.grad is None, n            # spatial_proj is never used


def _edge_head_case(vd, hidden, heads, counts, precision, seed):
    """EdgePredictor(vd, hidden, heads) on a ragged batch: forward and every gradient element against the fp64 oracle."""
    from models.EdgePredictor import EdgePredictor
    torch.manual_seed(seed)
    ep = EdgePredictor(vd, hidden, heads).to(dev())
    for sub in ep.modules():
        if isinstance(sub, torch.nn.Dropout):
            sub.p = 0.0
    ep.attention.dropout = 0.0
    ep.train()
    with torch.no_grad():
        for n, p_ in ep.named_parameters():
            if p_.dim() == 1:
                p_.add_(0.05 * torch.randn(p_.shape, generator=torch.Generator().manual_seed(len(n))).to(dev()))
    g = torch.Generator().manual_seed(7)
    vmax = max(counts)
    v = torch.randn(len(counts), vmax, vd, generator=g)
    vg = v.clone().to(dev()).requires_grad_()
    probs = ep.forward_ragged(vg, counts)
    cot = torch.randn(probs.shape, generator=g)
    (probs * cot.to(dev())).sum().backward()
    P = {"edge_predictor." + n: p_.detach().cpu().double().requires_grad_() for n, p_ in ep.named_parameters()}
    vr = v.double().requires_grad_()
    tot = 0.0
    for s_, c in enumerate(counts):
        pr, _ = oracle.edge_forward(P, vr[s_:s_ + 1, :c], num_heads=heads)
        assert H.elem_err(probs[s_, :pr.shape[1]].detach().cpu().numpy(), pr[0].detach().numpy(), 1e-6) < TOL_OUT
        if pr.shape[1] < probs.shape[1]:
            assert float(probs[s_, pr.shape[1]:].abs().max()) == 0.0          # padding exactly 0
        tot = tot + (pr[0] * cot[s_, :pr.shape[1]].double()).sum()
    tot.backward()
    tol = 1e-4 if precision == "fp32" else 3e-4
    assert H.elem_err(vg.grad.cpu().numpy(), vr.grad.numpy()) < tol
    for n, p_ in ep.named_parameters():
        ref = P["edge_predictor." + n].grad
        if ref is None:
            assert p_.grad is None, n
        else:
            assert H.elem_err(p_.grad.cpu().numpy(), ref.numpy()) < tol, (n, H.elem_err(p_.grad.cpu().numpy(), ref.numpy()))


@pytest.mark.parametrize("vd", [2, 4, 6])
def test_edge_head_other_vertex_dims_vs_oracle(vd, precision):
    """EdgePredictor(vertex_dim != 3) (reference models/EdgePredictor.py:19-31 is general in it; PointCloudToWireframe
    uses 3): forward and every gradient against the fp64 oracle on a ragged pair of samples.  vd <= 4 takes the
    coordinate columns as a low-rank GEMM epilogue, vd = 6 as a separate accumulate."""
    from models.EdgePredictor import EdgePredictor
    _edge_head_case(vd, 64, 2, [7, 4], precision, 40 + vd)
    with pytest.raises(ValueError):
        EdgePredictor(9, 64, 2)


@pytest.mark.parametrize("hidden,heads,counts", [(128, 8, [9, 3]), (256, 4, [12, 2, 7]), (384, 6, [5, 5]), (96, 3, [8, 6]),
                                                 (208, 4, [6, 4]), (256, 4, [50, 3]), (1024, 16, [6, 3])])
def test_edge_head_other_hidden_sizes_vs_oracle(hidden, heads, counts, precision):
    """EdgePredictor(hidden_dim, num_heads) other than (512, 8) (reference models/EdgePredictor.py:19): head widths 16 / 32 /
    52 on the VALU attention kernels, 64 on the MFMA ones, pair / LayerNorm kernels at row widths 96 ... 1024, and
    (256, 4, [50, 3]): 1,228 edge rows, enough for the split GEMMs of the edge MLP in bf16x3 mode."""
    _edge_head_case(3, hidden, heads, counts, precision, 60 + hidden)


@pytest.mark.parametrize("V", [0, 1])
def test_edge_degenerate_counts_raise_like_reference(V):
    from models.EdgePredictor import EdgePredictor
    ep = EdgePredictor(3, 64, 2).to(dev())
    with pytest.raises(IndexError):
        ep(torch.zeros(1, V, 3, device=dev()))


def test_small_vertex_vs_golden():
    from models.VertexPredictor import VertexPredictor
    gold = H.load_golden("small_vert")
    shapes = H.sub_shapes("vertex_predictor.", output_dim=16, max_vertices=5)
    vp = VertexPredictor(16, 5, 4).to(dev())
    vp.ensure_point_pool_proj(32, dev())
    load_arrays(vp, detgen.fill_state_dict(shapes, 5))
    g = torch.from_numpy(detgen.normalish("small_vert.g", (3, 16), 5)).to(dev()).requires_grad_()
    pf = torch.from_numpy(detgen.normalish("small_vert.pf", (3, 11, 16), 5)).to(dev()).requires_grad_()
    out = vp(g, pf)
    assert out["vertices"].shape == (3, 5, 3) and not out["vertices"].is_contiguous()
    assert H.rel_err(out["vertices"].detach().cpu().numpy(), gold["out.vertices"]) < TOL_OUT
    assert H.rel_err(out["existence_probabilities"].detach().cpu().numpy(), gold["out.exist"]) < TOL_OUT
    assert np.array_equal(out["actual_vertex_counts"].cpu().numpy(), gold["out.counts"])
    assert out["actual_vertex_counts"].dtype == torch.int64
    cv = torch.from_numpy(detgen.uniform("small_vert.cot.v", (3, 5, 3), -1, 1, 5)).to(dev())
    ce = torch.from_numpy(detgen.uniform("small_vert.cot.e", (3, 5), -1, 1, 5)).to(dev())
    ((out["vertices"] * cv).sum() + (out["existence_probabilities"] * ce).sum()).backward()
    assert H.rel_err(g.grad.cpu().numpy(), gold["grad.g"]) < TOL_GRAD
    assert H.rel_err(pf.grad.cpu().numpy(), gold["grad.pf"]) < TOL_GRAD
    bad = H.check_grad_summaries(gold, [(n, p.grad) for n, p in vp.named_parameters()], TOL_GRAD)
    assert not bad, bad
    out2 = vp(g.detach(), None)
    assert H.rel_err(out2["vertices"].detach().cpu().numpy(), gold["out.nopf.vertices"]) < TOL_OUT


def build_full(tag):
    from models.PointCloudToWireframe import PointCloudToWireframe
    gold = H.load_golden(tag)
    x, arrs, counts, V, seed, train = H.full_case_inputs(tag, gold)
    model = PointCloudToWireframe(8, V).to(dev())
    assert not hasattr(model.vertex_predictor, "point_pool_proj")          # lazy
    assert len(list(model.parameters())) == 78
    model.vertex_predictor.ensure_point_pool_proj(1024, dev())
    assert len(list(model.parameters())) == 80
    load_arrays(model, arrs)
    model.set_dropout(0.0)
    model.train(train)
    return gold, model, x, arrs, counts, V, seed, train


@pytest.mark.parametrize("tag", ["cfg1", "ragged", "evalmode"])
def test_full_model_vs_golden_and_oracle(tag, precision):
    gold, model, x, arrs, counts, V, seed, train = build_full(tag)
    xd = torch.from_numpy(x).to(dev())
    cd = counts.to(dev()) if counts is not None else None
    with torch.set_grad_enabled(train):
        out = model(xd, cd)
    # structure of the reference's return dict (PointCloudToWireframe.py:114-121)
    assert list(out.keys()) == ["vertices", "existence_probabilities", "edge_probs", "edge_indices",
                                "global_features", "actual_vertex_counts"]
    assert out["edge_probs"].dtype == torch.float32
    assert np.array_equal(out["actual_vertex_counts"].cpu().numpy(), gold["out.actual_vertex_counts"])
    lens = [len(e) for e in out["edge_indices"]]
    assert lens == gold["out.edge_index_lens"].tolist()
    flat = np.array([ij for e in out["edge_indices"] for ij in e], dtype=np.int64).reshape(-1, 2)
    assert np.array_equal(flat, gold["out.edge_indices_flat"])                   # bit-exact
    for k in ("vertices", "existence_probabilities", "edge_probs", "global_features"):
        got = out[k].detach().cpu().numpy()
        assert H.rel_err(got, gold["out." + k]) < TOL_OUT, k
        # element-wise: every value against ITS OWN magnitude (probabilities down to 1e-6, signed
        # outputs down to the tensor's rms — helpers.elem_err), not against the tensor maximum
        assert H.elem_err(got, gold["out." + k], H.OUT_FLOOR[k]) < TOL_OUT, k
    # padding must be exactly zero
    ep = out["edge_probs"].detach().cpu().numpy()
    for s, n in enumerate(lens):
        assert np.all(ep[s, n:] == 0.0)
    if train:
        cot = H.full_case_cotangents(tag, out, seed)
        loss = sum((out[k] * cot[k].to(dev())).sum() for k in cot)
        loss.backward()
        assert abs(loss.item() - float(gold["out.loss"])) < 1e-4 * max(1.0, abs(float(gold["out.loss"])))
        named = [(n, p.grad) for n, p in model.named_parameters()]
        bad = H.check_grad_summaries(gold, named, TOL_GRAD)
        assert not bad, bad
        for n, g in named:                           # spatial_proj gets no gradient at all
            if "spatial_proj" in n:
                assert g is None


def test_dropout_train_mode_runs_and_is_seeded():
    """train() with the reference's p=0.1: outputs differ from p=0, are finite,
    reproducible under torch.manual_seed, and backward is consistent with forward
    (finite differences on one weight)."""
    gold, model, x, arrs, counts, V, seed, train = build_full("ragged")
    model.set_dropout(0.1)
    model.train()
    xd, cd = torch.from_numpy(x).to(dev()), counts.to(dev())
    torch.manual_seed(7)
    a = model(xd, cd)["edge_probs"].detach().clone()
    torch.manual_seed(7)
    b = model(xd, cd)["edge_probs"].detach().clone()
    c = model(xd, cd)["edge_probs"].detach().clone()
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert torch.isfinite(a).all()
    model.set_dropout(0.0)
    d = model(xd, cd)["edge_probs"].detach()
    assert (a - d).abs().max() > 1e-3
    model.eval()
    with torch.no_grad():
        e1 = model(xd)["edge_probs"]
        e2 = model(xd)["edge_probs"]
    assert torch.equal(e1, e2)


def test_edge_head_dropout_gradcheck(precision):
    """Directional finite-difference check of EdgeFn backward WITH dropout masks on."""
    from models.EdgePredictor import EdgePredictor
    torch.manual_seed(3)
    ep = EdgePredictor(3, 64, 2).to(dev())
    ep.train()
    v = torch.randn(3, 6, 3, device=dev()).requires_grad_()
    counts = [6, 3, 6]                     # 15 + 3 + 15 = 33 edge rows... keep one ragged sample
    cot = torch.randn(3, 15, device=dev())

    def run(vv):
        torch.manual_seed(11)           # same dropout seed every call
        return (ep.forward_ragged(vv, counts) * cot).sum()

    loss = run(v)
    loss.backward()
    dirn = torch.randn_like(v)
    eps = 1e-2
    with torch.no_grad():
        lp = run(v + eps * dirn).double()
        lm = run(v - eps * dirn).double()
    fd = (lp - lm) / (2 * eps)
    an = (v.grad.double() * dirn.double()).sum()
    assert abs(fd - an) < 2e-2 * max(1.0, abs(an)), (fd.item(), an.item())
    w = ep.edge_mlp[4].weight
    dirw = torch.randn_like(w)
    an = (w.grad.double() * dirw.double()).sum()
    with torch.no_grad():
        w.add_(eps * dirw); lp = run(v).double()
        w.sub_(2 * eps * dirw); lm = run(v).double()
        w.add_(eps * dirw)
    fd = (lp - lm) / (2 * eps)
    assert abs(fd - an) < 2e-2 * max(1.0, abs(an)), (fd.item(), an.item())


@pytest.mark.parametrize("E,heads", [(128, 4), (256, 4), (512, 8)])      # head_dim 32 (VALU) and 64 (MFMA)
def test_attention_vs_torch_mha(E, heads):
    """wf3d_attn_fwd/bwd against torch's own MultiheadAttention math on ragged samples."""
    from wf3d import ops
    torch.manual_seed(0)
    counts = [5, 64, 130, 256, 2]
    meta = ops.EdgeMeta.get(counts, dev())
    qkv = torch.randn(meta.Rv, 3 * E, device=dev())
    dctx = torch.randn(meta.Rv, E, device=dev())
    ctx, lse = ops.attn_fwd(qkv, meta, E, heads)
    dqkv = ops.attn_bwd(qkv, dctx, ctx, lse, meta, E, heads)
    q64 = qkv.double().cpu().requires_grad_()
    outs = []
    off = 0
    hd = E // heads
    for c in counts:
        blk = q64[off:off + c]
        q, k, v = blk[:, :E], blk[:, E:2 * E], blk[:, 2 * E:]
        sp = lambda t: t.reshape(c, heads, hd).transpose(0, 1)     # noqa: E731
        s = (sp(q) / hd ** 0.5) @ sp(k).transpose(-1, -2)
        o = torch.softmax(s, -1) @ sp(v)
        outs.append(o.transpose(0, 1).reshape(c, E))
        off += c
    ref = torch.cat(outs)
    assert H.rel_err(ctx.cpu().numpy(), ref.detach().numpy()) < 2e-5
    (ref * dctx.double().cpu()).sum().backward()
    assert H.rel_err(dqkv.cpu().numpy(), q64.grad.numpy()) < 1e-4


@pytest.mark.parametrize("E,heads,drop", [(128, 4, 0.0), (512, 8, 0.0), (512, 8, 0.1)])
def test_attention_over_more_than_256_vertices(E, heads, drop):
    """Samples longer than the 256 key rows that fit the LDS next to V (max_vertices is data-dependent in the reference,
    train.py:37): keys / queries are walked in chunks with the running (lse, ctx) merged per chunk.  Against torch's
    MultiheadAttention math in fp64; with dropout the masks of forward and backward must be the same ones (the check is
    the finite-difference identity d<ctx, c>/d qkv along a random direction)."""
    from wf3d import ops
    torch.manual_seed(1)
    counts = [300, 513, 7, 257]
    meta = ops.EdgeMeta.get(counts, dev())
    qkv = torch.randn(meta.Rv, 3 * E, device=dev())
    dctx = torch.randn(meta.Rv, E, device=dev())
    ctx, lse = ops.attn_fwd(qkv, meta, E, heads, drop, 99)
    dqkv = ops.attn_bwd(qkv, dctx, ctx, lse, meta, E, heads, drop, 99)
    if drop == 0.0:
        q64 = qkv.double().cpu().requires_grad_()
        outs, off, hd = [], 0, E // heads
        for c in counts:
            blk = q64[off:off + c]
            q, k, v = blk[:, :E], blk[:, E:2 * E], blk[:, 2 * E:]
            sp = lambda t: t.reshape(c, heads, hd).transpose(0, 1)     # noqa: E731
            o = torch.softmax((sp(q) / hd ** 0.5) @ sp(k).transpose(-1, -2), -1) @ sp(v)
            outs.append(o.transpose(0, 1).reshape(c, E))
            off += c
        ref = torch.cat(outs)
        assert H.rel_err(ctx.cpu().numpy(), ref.detach().numpy()) < 2e-5
        (ref * dctx.double().cpu()).sum().backward()
        assert H.rel_err(dqkv.cpu().numpy(), q64.grad.numpy()) < 1e-4
    else:
        ctx2, _ = ops.attn_fwd(qkv, meta, E, heads, drop, 99)
        assert torch.equal(ctx, ctx2)                                   # counter-based masks: reproducible
        d = torch.randn_like(qkv)
        eps = 1e-2
        lp = (ops.attn_fwd(qkv + eps * d, meta, E, heads, drop, 99)[0].double() * dctx.double()).sum()
        lm = (ops.attn_fwd(qkv - eps * d, meta, E, heads, drop, 99)[0].double() * dctx.double()).sum()
        fd, an = float((lp - lm) / (2 * eps)), float((dqkv.double() * d.double()).sum())
        assert abs(fd - an) < 2e-3 * max(1.0, abs(an)), (fd, an)


def test_split_mode_is_active_and_close_to_fp32():
    """At cfg1 size the default mode must take the split-GEMM path (kernel really used) and
    agree with the fp32 mode to ~1e-5 on outputs."""
    from wf3d import config, ops
    gold, model, x, arrs, counts, V, seed, train = build_full("cfg1")
    xd, cd = torch.from_numpy(x).to(dev()), counts.to(dev())
    calls = {"nt": 0, "tn": 0}
    orig, orig_tn = ops.gemm_split, ops.gemm_split_tn

    def spy(*a, **k):
        calls["nt"] += 1
        return orig(*a, **k)

    def spy_tn(*a, **k):
        calls["tn"] += 1
        return orig_tn(*a, **k)

    ops.gemm_split, ops.gemm_split_tn = spy, spy_tn
    try:
        assert config.precision() == "bf16x3"
        out_s = model(xd, cd)
        sum(out_s[k].sum() for k in ("vertices", "edge_probs")).backward()
    finally:
        ops.gemm_split, ops.gemm_split_tn = orig, orig_tn
    # layers 2..5 of the per-point MLP: 4 forward + 4 dgrad (NT form) and 4 wgrad (TN form) split GEMMs
    assert calls == {"nt": 8, "tn": 4}
    config.set_precision("fp32")
    try:
        out_f = model(xd, cd)
    finally:
        config.set_precision("bf16x3")
    for k in ("vertices", "existence_probabilities", "edge_probs", "global_features"):
        e = H.rel_err(out_s[k].detach().cpu().numpy(), out_f[k].detach().cpu().numpy())
        assert e < 5e-5, (k, e)


# ---------------------------------------------------------------------------
# cfg5's edge head (max_vertices = 256: E = 32,640 pair rows per sample) against the oracle
# ---------------------------------------------------------------------------
def test_edge_head_v256_vs_oracle(precision):
    """EdgePredictor at V=256 (reference models/EdgePredictor.py:117-138: pair features, the two wide
    edge-MLP layers, MFMA attention over 256 keys) against the CPU oracle.  The edge head has no
    ReLU / arg-max (LayerNorm, GELU, softmax, sigmoid only), so outputs AND gradients are held
    element-wise with no decision freezing: outputs 1e-4 of their own value, gradients 1e-4 of
    max(|g|, rms(g)) against the oracle in fp64."""
    from models.EdgePredictor import EdgePredictor
    torch.manual_seed(5)
    ep = EdgePredictor(3, 512, 8).to(dev())
    for sub in ep.modules():
        if isinstance(sub, torch.nn.Dropout):
            sub.p = 0.0
    ep.attention.dropout = 0.0
    ep.train()
    gen = torch.Generator().manual_seed(6)
    v = torch.randn(2, 256, 3, generator=gen)
    counts = [256, 131]
    vd = v.to(dev()).requires_grad_()
    probs = ep.forward_ragged(vd, counts)
    assert probs.shape == (2, 32640)
    cot = torch.randn(probs.shape, generator=gen)
    (probs * cot.to(dev())).sum().backward()
    P = {"edge_predictor." + k: t.detach().cpu().double().requires_grad_() for k, t in ep.state_dict().items()}
    v64 = v.double().requires_grad_()
    loss = 0.0
    for s, c in enumerate(counts):
        p_ref, idx = oracle.edge_forward(P, v64[s:s + 1, :c], 8)
        e = c * (c - 1) // 2
        got = probs[s, :e].detach().cpu().numpy()
        assert H.elem_err(got, p_ref[0].detach().numpy(), 1e-6) < TOL_OUT, s
        assert float(probs[s, e:].abs().max()) == 0.0 if e < probs.shape[1] else True
        loss = loss + (p_ref[0] * cot[s, :e].double()).sum()
    loss.backward()
    tol = 1e-4 if precision == "fp32" else 2e-4          # bf16x3: measured 7.7e-5 here, 1.1e-4 at worst
    assert H.elem_err(vd.grad.cpu().numpy(), v64.grad.numpy()) < tol
    worst = 0.0
    for n, p in ep.named_parameters():
        ref = P["edge_predictor." + n].grad
        if ref is None:
            assert p.grad is None, n
            continue
        if n == "attention.in_proj_bias":
            # the key third of this gradient is analytically zero (softmax is shift-invariant): both
            # sides hold rounding noise there; compare the query and value thirds
            g, r = p.grad.cpu().numpy(), ref.numpy()
            e = max(H.elem_err(g[:512], r[:512]), H.elem_err(g[1024:], r[1024:]))
            assert np.abs(g[512:1024]).max() < 1e-4 * np.abs(r).max()
        else:
            e = H.elem_err(p.grad.cpu().numpy(), ref.numpy())
        worst = max(worst, e)
        assert e < tol, (n, e)
    print(f"V=256 edge head [{precision}]: worst element-wise gradient error {worst:.2e}")


@pytest.mark.parametrize("V", [256, 300])
def test_full_model_v256_tiny_cloud_vs_oracle(precision, V):
    """Whole model at max_vertices = 256 — and 300, beyond the attention kernels' LDS-resident key block (chunked keys) —
    on a tiny cloud (B=2, N=64; the oracle needs ~1 s per sample)."""
    from models.PointCloudToWireframe import PointCloudToWireframe
    torch.manual_seed(8)
    model = PointCloudToWireframe(8, V).to(dev()).set_dropout(0.0)
    model.train()
    gen = torch.Generator().manual_seed(9)
    x = torch.randn(2, 64, 8, generator=gen)
    counts = torch.tensor([V, 77])
    out = model(x.to(dev()), counts.to(dev()))
    P = oracle.params_from_module(model)
    with torch.no_grad():
        ref = oracle.model_forward(P, x, counts, V, training=True)
    assert out["edge_indices"] == ref["edge_indices"]
    assert out["edge_probs"].shape == ref["edge_probs"].shape == (2, V * (V - 1) // 2)
    for k, (a, b) in H.out_errs(out, ref).items():
        assert a < TOL_OUT and b < TOL_OUT, (k, a, b)
    assert float(out["edge_probs"][1, 77 * 76 // 2:].abs().max()) == 0.0


def test_fresh_count_tensors_are_read_every_call():
    """Two batches with DIFFERENT freshly allocated count tensors back to back (the caching allocator hands
    the second one the first one's address): edge rows must follow the second batch's counts
    (reference semantics: PointCloudToWireframe.py:79-81 reads the counts on every forward)."""
    from models.PointCloudToWireframe import PointCloudToWireframe
    torch.manual_seed(2)
    model = PointCloudToWireframe(8, 12).to(dev()).set_dropout(0.0)
    model.train()
    x = torch.randn(3, 128, 8, device=dev())

    def step(vals):
        counts = torch.tensor(vals).to(dev())           # created and freed inside: same address next time
        out = model(x, counts)
        return counts.data_ptr(), [len(e) for e in out["edge_indices"]], out["edge_probs"].shape[1]

    p1, l1, w1 = step([12, 3, 7])
    p2, l2, w2 = step([4, 12, 2])
    assert l1 == [66, 3, 21] and w1 == 66
    assert l2 == [6, 66, 1] and w2 == 66
    p3, l3, w3 = step([5, 5, 5])
    assert l3 == [10, 10, 10] and w3 == 10
    # same tensor object, modified in place: the version counter invalidates the cached host copy
    c = torch.tensor([12, 3, 7]).to(dev())
    assert [len(e) for e in model(x, c)["edge_indices"]] == [66, 3, 21]
    c.copy_(torch.tensor([2, 2, 9]))
    assert [len(e) for e in model(x, c)["edge_indices"]] == [1, 1, 36]
    assert [len(e) for e in model(x, c)["edge_indices"]] == [1, 1, 36]       # unchanged tensor: cached path


def test_second_backward_raises_a_clear_error():
    from models.PointCloudToWireframe import PointCloudToWireframe
    torch.manual_seed(2)
    model = PointCloudToWireframe(8, 6).to(dev()).set_dropout(0.0)
    model.train()
    out = model(torch.randn(2, 64, 8, device=dev()), torch.tensor([6, 4]))
    loss = out["vertices"].sum() + out["edge_probs"].sum()
    loss.backward(retain_graph=True)
    with pytest.raises(RuntimeError, match="second time"):
        loss.backward()


def test_retain_graph_is_served_when_the_saved_activations_are_kept():
    """wf3d.config.RETAIN_SAVED = True: the stages keep their saved activations, so backward(retain_graph=True) followed by
    a second backward works as with the reference's autograd — the second pass accumulates the same gradients again."""
    from models.PointCloudToWireframe import PointCloudToWireframe
    from wf3d import config
    torch.manual_seed(2)
    model = PointCloudToWireframe(8, 6).to(dev()).set_dropout(0.0)
    model.train()
    old = config.RETAIN_SAVED
    config.RETAIN_SAVED = True
    try:
        out = model(torch.randn(2, 64, 8, device=dev()), torch.tensor([6, 4]))
        loss = out["vertices"].sum() + out["edge_probs"].sum()
        loss.backward(retain_graph=True)
        g1 = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
        loss.backward()
        for n, p in model.named_parameters():
            if p.grad is not None:
                assert torch.allclose(p.grad, 2 * g1[n], rtol=1e-6, atol=1e-9), n
    finally:
        config.RETAIN_SAVED = old


def test_inplace_parameter_update_between_forward_and_backward_is_reported():
    """The stages keep their parameters as plain references; an optimizer step (or any in-place write) between the
    forward and its backward must raise, as autograd does for saved tensors — not differentiate against new values."""
    from models.PointCloudToWireframe import PointCloudToWireframe
    from wf3d.optim import ClipAdam
    torch.manual_seed(3)
    model = PointCloudToWireframe(8, 6).to(dev()).set_dropout(0.0)
    model.train()
    x = torch.randn(2, 40, 8, device=dev())
    counts = torch.tensor([6, 3], device=dev())
    out = model(x, counts)
    with torch.no_grad():
        model.edge_predictor.edge_mlp[0].weight.mul_(1.0)            # in-place write, values unchanged
    with pytest.raises(RuntimeError, match="modified in place"):
        out["edge_probs"].sum().backward()
    # the fused optimizer writes through raw pointers and must bump the counters itself
    out = model(x, counts)
    (out["edge_probs"].sum() + out["vertices"].sum() + out["existence_probabilities"].sum()).backward()
    opt = ClipAdam(model.parameters(), lr=1e-3, max_norm=1.0)
    out2 = model(x, counts)
    opt.step()
    with pytest.raises(RuntimeError, match="modified in place"):
        out2["vertices"].sum().backward()


def test_input_cloud_gradient_matches_oracle(precision):
    """A cloud with requires_grad (the reference supports it, train.py never asks): dx against the fp64 oracle run under
    the kernels' own ReLU / arg-max decisions, element-wise."""
    from models.PointCloudToWireframe import PointCloudToWireframe
    torch.manual_seed(12)
    V = 5
    model = PointCloudToWireframe(8, V).to(dev()).set_dropout(0.0)
    model.train()
    for seed in range(40, 72):
        gen = torch.Generator().manual_seed(seed)
        x = torch.randn(2, 48, 8, generator=gen)
        xd = x.to(dev()).requires_grad_()
        counts = torch.tensor([5, 3])
        out = model(xd, counts.to(dev()))
        frozen, n_border = H.capture_decisions(out, model)
        if n_border == 0:
            break
    else:
        pytest.fail("no borderline-free input found")
    cot = {k: torch.randn(out[k].shape, generator=gen) for k in ("vertices", "existence_probabilities", "edge_probs")}
    sum((out[k] * cot[k].to(dev())).sum() for k in cot).backward()
    assert xd.grad is not None and xd.grad.shape == x.shape
    P = oracle.params_from_module(model, dtype=torch.float64)
    x64 = x.double().requires_grad_()
    ref = oracle.model_forward(P, x64, counts, V, training=True, frozen=frozen)
    sum((ref[k] * cot[k].double()).sum() for k in cot).backward()
    e = H.elem_err(xd.grad.cpu().numpy(), x64.grad.numpy())
    assert e < (1e-4 if precision == "fp32" else 2e-4), e


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
@pytest.mark.parametrize("hidden,out", [([256, 768], 768), ([512], 96), ([264, 520], 264), ([1024, 1536], 1280), ([64, 128, 256, 128, 64], 32), ([], 40)])
def test_encoder_other_widths_vs_oracle(precision, hidden, out):
    """PointNetEncoder is general in hidden_dims / output_dim (reference models/PointNetEncoder.py:19-47): widths that are
    not the model's — slots of the row passes partly filled, channel counts whose quarter does not divide 256, layer
    widths the persistent GEMM does not tile — against the fp64 oracle, rows enough for the split path (3,072)."""
    from models.PointNetEncoder import PointNetEncoder
    from wf3d import config
    old = config.precision()
    config.set_precision(precision)
    try:
        torch.manual_seed(5)
        enc = PointNetEncoder(8, hidden, out).to(dev())
        with torch.no_grad():
            for n, p in enc.named_parameters():
                if p.dim() == 1:
                    p.add_(0.05 * torch.randn(p.shape, generator=torch.Generator().manual_seed(len(n))).to(dev()))
        g0 = torch.Generator().manual_seed(9)
        x = torch.randn(3, 1024, 8, generator=g0)
        x[:, ::9] = 0.0
        cg, cp = torch.randn(3, out, generator=g0), torch.randn(3, 1024, out, generator=g0) * 0.01
        g, pf = enc(x.to(dev()))
        ((g * cg.to(dev())).sum() + (pf * cp.to(dev())).sum()).backward()
        P = oracle.params_from_module(enc, dtype=torch.float64)
        P = {"encoder." + k: v for k, v in P.items()}
        rg, rpf = oracle.encoder_forward(P, x.double())
        ((rg * cg.double()).sum() + (rpf * cp.double()).sum()).backward()
        assert H.rel_err(g.detach().cpu().numpy(), rg.detach().numpy()) < TOL_OUT
        assert H.rel_err(pf.detach().cpu().numpy(), rpf.detach().numpy()) < TOL_OUT
        for n, p in enc.named_parameters():
            ref = P["encoder." + n].grad
            assert ref is not None and p.grad is not None, n
            # not decision-frozen: ReLU decisions at |y| below the arithmetic's forward error differ from fp64's (fp32: under
            # one element per layer here; bf16x3: dozens), and a flipped element on a row that is some channel's arg-max
            # carries that channel's whole pooled cotangent.  So: every element for fp32 (measured 6e-3 at worst), the L2
            # norm for bf16x3 (a lane pair writing the wrong 8 of 520 columns would be 0.12).
            a, b = p.grad.double().cpu(), ref
            if precision == "fp32":
                assert H.rel_err(a.numpy(), b.numpy()) < 2e-2, (n, H.rel_err(a.numpy(), b.numpy()))
            else:
                l2 = float((a - b).norm() / b.norm().clamp_min(1e-30))
                assert l2 < 3e-2, (n, l2)
    finally:
        config.set_precision(old)


@pytest.mark.parametrize("gdim,V,B,npts", [(384, 10, 5, 17), (512, 100, 3, 9), (200, 7, 33, 12), (520, 3, 2, 33), (64, 40, 32, 5), (1024, 12, 4, 8)])
def test_vertex_head_other_sizes_vs_oracle(gdim, V, B, npts):
    """VertexPredictor(global_feature_dim, max_vertices) other than (512, 64) (reference models/VertexPredictor.py:19):
    input widths that are not multiples of 512, the skinny kernels (B <= 32) and the generic GEMM path (B = 33),
    against the fp64 oracle — outputs and every gradient, relative to each tensor's scale."""
    from models.VertexPredictor import VertexPredictor
    torch.manual_seed(gdim + V)
    vp = VertexPredictor(gdim, V, 4).to(dev())
    vp.ensure_point_pool_proj(2 * gdim, dev())
    with torch.no_grad():
        for n, p in vp.named_parameters():
            if p.dim() == 1:
                p.add_(0.05 * torch.randn(p.shape, generator=torch.Generator().manual_seed(len(n))).to(dev()))
    g0 = torch.Generator().manual_seed(3)
    g = torch.randn(B, gdim, generator=g0)
    pf = torch.randn(B, npts, gdim, generator=g0)
    gg, pg = g.clone().to(dev()).requires_grad_(), pf.clone().to(dev()).requires_grad_()
    out = vp(gg, pg)
    cv, ce = torch.randn(B, V, 3, generator=g0), torch.randn(B, V, generator=g0)
    ((out["vertices"] * cv.to(dev())).sum() + (out["existence_probabilities"] * ce.to(dev())).sum()).backward()
    P = {"vertex_predictor." + n: p.detach().cpu().double().requires_grad_() for n, p in vp.named_parameters()}
    gr, pr = g.double().requires_grad_(), pf.double().requires_grad_()
    ref = oracle.vertex_forward(P, gr, pr, V, 4)
    ((ref["vertices"] * cv.double()).sum() + (ref["existence_probabilities"] * ce.double()).sum()).backward()
    assert H.rel_err(out["vertices"].detach().cpu().numpy(), ref["vertices"].detach().numpy()) < TOL_OUT
    assert H.rel_err(out["existence_probabilities"].detach().cpu().numpy(), ref["existence_probabilities"].detach().numpy()) < TOL_OUT
    assert np.array_equal(out["actual_vertex_counts"].cpu().numpy(), ref["actual_vertex_counts"].numpy())
    assert H.rel_err(gg.grad.cpu().numpy(), gr.grad.numpy()) < TOL_GRAD
    assert H.rel_err(pg.grad.cpu().numpy(), pr.grad.numpy()) < TOL_GRAD
    for n, p in vp.named_parameters():
        r = P["vertex_predictor." + n].grad
        assert r is not None and p.grad is not None, n
        assert H.rel_err(p.grad.cpu().numpy(), r.numpy()) < TOL_GRAD, (n, H.rel_err(p.grad.cpu().numpy(), r.numpy()))


def test_input_layouts_and_dtypes_give_the_same_result():
    """Strided / non-fp32 point clouds and count tensors in other integer types and devices are accepted and mean the same
    thing (the reference's modules take whatever torch ops take)."""
    from models.PointCloudToWireframe import PointCloudToWireframe
    torch.manual_seed(11)
    m = PointCloudToWireframe(8, 6).to(dev())
    m.vertex_predictor.ensure_point_pool_proj(1024, dev())
    m.set_dropout(0.0)
    m.train()
    g = torch.Generator().manual_seed(2)
    big = torch.randn(3, 2 * 300, 8, generator=g).to(dev())
    x = big[:, ::2, :]                                       # strided view
    counts = torch.tensor([6, 2, 4])
    for xv, cv in ((x, counts.to(dev())), (x.double(), counts), (x.contiguous().half(), counts.int().to(dev())),
                   (x.permute(1, 0, 2).contiguous().permute(1, 0, 2), counts.to(dev()).to(torch.int16))):
        out = m(xv, cv)
        ref = m(xv.float().contiguous(), counts.to(dev()))       # the same values as a plain fp32 tensor and int64 counts
        for k in ("vertices", "existence_probabilities", "edge_probs", "global_features"):
            assert torch.equal(out[k], ref[k]), k
        assert out["edge_indices"] == ref["edge_indices"]


def test_count_tensor_shorter_or_longer_than_the_batch():
    """The reference reads target_vertex_counts[i] for i < batch: too few entries is its IndexError, extra ones are ignored."""
    from models.PointCloudToWireframe import PointCloudToWireframe
    torch.manual_seed(12)
    m = PointCloudToWireframe(8, 6).to(dev())
    m.set_dropout(0.0)
    m.train()
    x = torch.randn(2, 64, 8, device=dev())
    with pytest.raises(IndexError, match="out of bounds"):
        m(x, torch.tensor([5], device=dev()))
    a, b = m(x, torch.tensor([5, 3, 6, 2], device=dev())), m(x, torch.tensor([5, 3], device=dev()))
    assert torch.equal(a["edge_probs"], b["edge_probs"]) and a["edge_indices"] == b["edge_indices"]


def test_forward_ragged_counts_beyond_the_padded_width_are_cut():
    from models.EdgePredictor import EdgePredictor
    torch.manual_seed(13)
    ep = EdgePredictor(3, 64, 2).to(dev()).eval()
    v = torch.randn(2, 5, 3, device=dev())
    a, b = ep.forward_ragged(v, [9, 3]), ep.forward_ragged(v, [5, 3])
    assert torch.equal(a, b)
    with pytest.raises(ValueError, match="counts for a batch"):
        ep.forward_ragged(v, [5])


@pytest.mark.parametrize("hidden,heads", [(120, 5), (72, 3), (8, 1), (520, 8), (100, 4), (64, 3)])
def test_edge_head_sizes_outside_the_kernels_granularity_are_refused_at_construction(hidden, heads):
    from models.EdgePredictor import EdgePredictor
    with pytest.raises(ValueError, match="hidden_dim"):
        EdgePredictor(3, hidden, heads)


@pytest.mark.parametrize("hidden,out", [([30, 50], 22), ([33], 18), ([6], 2), ([512, 8192], 64), ([64], 17)])
def test_encoder_sizes_outside_the_kernels_granularity_are_refused_at_construction(hidden, out):
    from models.PointNetEncoder import PointNetEncoder
    with pytest.raises(ValueError, match="hidden"):
        PointNetEncoder(8, hidden, out)
    PointNetEncoder(8, [36, 20], 12)                  # multiples of 4 / even: fine
