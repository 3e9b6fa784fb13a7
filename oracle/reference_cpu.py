"""CPU oracle for the PointNet-encoder -> vertex head -> edge head path.

TEST INFRASTRUCTURE ONLY.  This is a from-scratch, plain-torch *functional*
restatement of the arithmetic of the reference's hot path.  It is what the HIP
kernels are checked against and what bench.py times as `cpu_baseline`
(kind "port").  The product package never imports it: only tests/,
tests/golden/make_golden.py, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may.

Parity pin: tests/test_oracle_golden.py checks every function here against the
fixtures in tests/golden/*.npz, which tests/golden/make_golden.py produced by
importing the reference's own model classes from /root/reference on CPU
(torch 2.10.0), and tests/test_oracle_vs_reference.py re-checks directly
against the imported reference whenever /root/reference is present.

All parameters are addressed by the reference's state_dict names, e.g.
``encoder.mlp.4.weight`` (reference layouts: models/PointNetEncoder.py:35-65,
models/VertexPredictor.py:27-61, models/EdgePredictor.py:31-68).
Dropout is taken as p = 0 everywhere (the reference's p = 0.1 dropouts use the
global torch RNG and are not reproducible by any other implementation;
SURVEY.md §9 Q3).
"""
import math

import torch
import torch.nn.functional as F

LN_EPS = 1e-5  # nn.LayerNorm default, used by every LayerNorm on the path


def _lin(P, name, x):
    return F.linear(x, P[name + ".weight"], P[name + ".bias"])


def _ln(P, name, x):
    w = P[name + ".weight"]
    return F.layer_norm(x, (w.shape[0],), w, P[name + ".bias"], LN_EPS)


def _relu_ln(P, name, u, frozen=None):
    """relu(LayerNorm(u)).  With `frozen` (decision-frozen mode, tests/test_frozen_grad_gpu.py) the
    ReLU's 0/1 decision of LayerNorm `name` is not re-taken from this run's values but read from
    frozen["relu"][name] (a bool tensor recorded from another run of the same network), so that the
    result is a smooth function of the arithmetic: relu(v) -> v * mask."""
    v = _ln(P, name, u)
    if frozen is not None and name in frozen.get("relu", {}):
        return v * frozen["relu"][name].to(v.dtype)
    return torch.relu(v)


def _max_over_points(pf, frozen, key, fill_dead=False):
    """max over dim 1 of pf [B,N,C]; decision-frozen mode gathers the recorded arg-max rows
    frozen["argmax"][key] ([B,C] int64, -1 = no valid row -> 0) instead of re-deciding them."""
    idx = frozen["argmax"][key]
    got = torch.gather(pf, 1, idx.clamp(min=0).unsqueeze(1)).squeeze(1)
    return torch.where(idx >= 0, got, torch.zeros_like(got))


def _count_blocks(P, prefix, stride):
    n = 0
    while f"{prefix}{stride * n}.weight" in P:
        n += 1
    return n


# --------------------------------------------------------------------------
# PointNetEncoder.forward  (reference models/PointNetEncoder.py:67-118)
# --------------------------------------------------------------------------
def encoder_point_mlp(P, x2d, prefix="encoder.", frozen=None):
    """Per-point shared MLP (reference PointNetEncoder.py:35-45,94): blocks of
    Linear -> LayerNorm -> ReLU at Sequential indices 4i, 4i+1, closed by a
    bare Linear at index 4*n_hidden."""
    mp = prefix + "mlp."
    n_lin = _count_blocks(P, mp, 4)           # Linears sit at 0,4,8,...
    h = x2d
    for i in range(n_lin - 1):
        h = _relu_ln(P, f"{mp}{4 * i + 1}", _lin(P, f"{mp}{4 * i}", h), frozen)
    return _lin(P, f"{mp}{4 * (n_lin - 1)}", h)


def encoder_pools(x, pf, frozen=None):
    """Mask-aware pools (reference PointNetEncoder.py:85-86,103-111)."""
    valid = x.detach().abs().sum(-1) > 1e-9                       # [B,N]
    cnt = valid.sum(1, keepdim=True).clamp(min=1).to(pf.dtype)     # [B,1]
    avg = (pf * valid.unsqueeze(-1)).sum(1) / cnt
    if frozen is not None and "enc_masked" in frozen.get("argmax", {}):
        return _max_over_points(pf, frozen, "enc_masked"), avg
    neg = torch.full_like(pf, float("-inf"))
    mx = torch.where(valid.unsqueeze(-1), pf, neg).max(dim=1).values
    mx = torch.where(torch.isfinite(mx), mx, torch.zeros_like(mx))
    return mx, avg


def encoder_fusion(P, pooled, prefix="encoder.", frozen=None):
    """feature_fusion (reference PointNetEncoder.py:57-65): Linear,LN,ReLU,
    Linear,LN,ReLU,Linear at Sequential indices 0,1,3,4,6."""
    fp = prefix + "feature_fusion."
    h = _relu_ln(P, fp + "1", _lin(P, fp + "0", pooled), frozen)
    h = _relu_ln(P, fp + "4", _lin(P, fp + "3", h), frozen)
    return _lin(P, fp + "6", h)


def encoder_forward(P, x, prefix="encoder.", frozen=None):
    B, N, D = x.shape
    pf = encoder_point_mlp(P, x.reshape(B * N, D), prefix, frozen).reshape(B, N, -1)
    mx, avg = encoder_pools(x, pf, frozen)
    g = encoder_fusion(P, torch.cat([mx, avg], dim=1), prefix, frozen)     # max first (:115)
    return g, pf


# --------------------------------------------------------------------------
# VertexPredictor.forward  (reference models/VertexPredictor.py:63-133)
# --------------------------------------------------------------------------
def vertex_forward(P, g, pf, max_vertices, vertex_dim=4, prefix="vertex_predictor.", frozen=None):
    if pf is not None:
        # UNMASKED mean/max over all N points, mean first (:86-88), through the
        # lazily created point_pool_proj (:94-99), added to the global vector.
        if frozen is not None and "vert_unmasked" in frozen.get("argmax", {}):
            umax = _max_over_points(pf, frozen, "vert_unmasked")
        else:
            umax = pf.max(dim=1).values
        pooled = torch.cat([pf.mean(dim=1), umax], dim=1)
        e = g + _lin(P, prefix + "point_pool_proj", pooled)
    else:
        e = g
    a = _relu_ln(P, prefix + "vertex_mlp1.1", _lin(P, prefix + "vertex_mlp1.0", e), frozen)
    b = _relu_ln(P, prefix + "vertex_mlp2.1", _lin(P, prefix + "vertex_mlp2.0", a), frozen)
    c = _relu_ln(P, prefix + "vertex_mlp3.1", _lin(P, prefix + "vertex_mlp3.0", b), frozen)
    c = c + _lin(P, prefix + "residual_proj1", e)                 # residual after ReLU (:110)
    d = _relu_ln(P, prefix + "vertex_mlp4.1", _lin(P, prefix + "vertex_mlp4.0", c), frozen)
    d = d + _lin(P, prefix + "residual_proj2", e)
    o = _lin(P, prefix + "final_layer", d).reshape(g.shape[0], max_vertices, vertex_dim)
    coords = o[:, :, :3]
    exist = torch.sigmoid(o[:, :, 3])
    counts = (exist > 0.5).sum(dim=1)
    return {"vertices": coords, "existence_probabilities": exist,
            "actual_vertex_counts": counts}


# --------------------------------------------------------------------------
# EdgePredictor  (reference models/EdgePredictor.py:70-140)
# --------------------------------------------------------------------------
def edge_index_pairs(v):
    """All (i, j), i < j, lexicographic — the order of the reference's nested
    loop (EdgePredictor.py:83-86).  Returned as a Python list of [i, j]."""
    return [[i, j] for i in range(v) for j in range(i + 1, v)]


def edge_vertex_embed(P, c, prefix="edge_predictor."):
    vp = prefix + "vertex_proj."
    h = F.gelu(_ln(P, vp + "1", _lin(P, vp + "0", c)))            # erf GELU
    return _ln(P, vp + "4", _lin(P, vp + "3", h))                 # Dropout(.1) -> identity


def edge_self_attention(P, f, num_heads, prefix="edge_predictor."):
    """nn.MultiheadAttention(batch_first) self-attention, need_weights path
    (torch nn/functional.py multi_head_attention_forward): packed in_proj,
    q scaled by 1/sqrt(head_dim) BEFORE the product, softmax over keys,
    heads concatenated, out_proj."""
    ap = prefix + "attention."
    Bs, V, E = f.shape
    hd = E // num_heads
    qkv = F.linear(f, P[ap + "in_proj_weight"], P[ap + "in_proj_bias"])
    q, k, v = qkv.split(E, dim=-1)

    def heads(t):
        return t.reshape(Bs, V, num_heads, hd).transpose(1, 2)     # [Bs,H,V,hd]

    q = heads(q) * math.sqrt(1.0 / hd)
    s = q @ heads(k).transpose(-1, -2)
    p = torch.softmax(s, dim=-1)
    ctx = (p @ heads(v)).transpose(1, 2).reshape(Bs, V, E)
    return F.linear(ctx, P[ap + "out_proj.weight"], P[ap + "out_proj.bias"])


def edge_pair_mlp(P, z, prefix="edge_predictor."):
    ep = prefix + "edge_mlp."
    h = F.gelu(_ln(P, ep + "1", _lin(P, ep + "0", z)))
    h = F.gelu(_ln(P, ep + "5", _lin(P, ep + "4", h)))
    h = F.gelu(_lin(P, ep + "8", h))
    return _lin(P, ep + "10", h)


def edge_forward(P, verts, num_heads=8, prefix="edge_predictor."):
    """verts [Bs, V, 3] -> (probs [Bs, E], list of [i, j]).  V <= 1 raises
    IndexError like the reference (EdgePredictor.py:118)."""
    Bs, V, _ = verts.shape
    f = edge_vertex_embed(P, verts, prefix)
    f = f + edge_self_attention(P, f, num_heads, prefix)
    pairs = edge_index_pairs(V)
    if not pairs:
        raise IndexError("too many indices for tensor of dimension 1")
    idx = torch.tensor(pairs, dtype=torch.long, device=verts.device)
    i, j = idx[:, 0], idx[:, 1]
    ci, cj = verts[:, i, :], verts[:, j, :]
    dist = torch.linalg.vector_norm(ci - cj, dim=-1, keepdim=True)
    z = torch.cat([f[:, i, :], f[:, j, :], ci, cj, dist], dim=-1)   # 512|512|3|3|1
    logits = edge_pair_mlp(P, z.reshape(-1, z.shape[-1]), prefix)
    return torch.sigmoid(logits).reshape(Bs, -1), pairs


# --------------------------------------------------------------------------
# PointCloudToWireframe.forward  (reference models/PointCloudToWireframe.py:43-121)
# --------------------------------------------------------------------------
def model_forward(P, x, target_vertex_counts, max_vertices, training=True, num_heads=8, frozen=None):
    """frozen: optional {"relu": {LayerNorm name: bool mask}, "argmax": {"enc_masked"|"vert_unmasked": [B,C]}}
    — the piecewise-constant decisions of another run, see _relu_ln."""
    g, pf = encoder_forward(P, x, frozen=frozen)
    vo = vertex_forward(P, g, pf, max_vertices, frozen=frozen)
    verts = vo["vertices"]
    B = verts.shape[0]
    if training and target_vertex_counts is not None:
        counts = [int(c) for c in target_vertex_counts.tolist()]
    else:
        counts = [int(c) for c in vo["actual_vertex_counts"].tolist()]
    probs, indices = [], []
    for s in range(B):
        p, idx = edge_forward(P, verts[s:s + 1, :counts[s], :], num_heads)
        probs.append(p[0])
        indices.append(idx)
    max_e = max((len(p) for p in probs), default=0)
    # always fp32 in the reference (:107); the fp64 decision-frozen runs keep their own dtype
    padded = torch.zeros(B, max_e, device=verts.device, dtype=torch.float32 if frozen is None else verts.dtype)
    for s, p in enumerate(probs):
        if len(p) > 0:
            padded[s, :len(p)] = p
    return {"vertices": verts,
            "existence_probabilities": vo["existence_probabilities"],
            "edge_probs": padded,
            "edge_indices": indices,
            "global_features": g,
            "actual_vertex_counts": vo["actual_vertex_counts"]}


# --------------------------------------------------------------------------
# helpers for tests / bench
# --------------------------------------------------------------------------
def params_from_numpy(arrs, requires_grad=True, dtype=torch.float32):
    return {k: torch.tensor(v, dtype=dtype).requires_grad_(requires_grad)
            for k, v in arrs.items()}


def params_from_module(module, requires_grad=True, dtype=torch.float32):
    """Detached CPU copies of a module's state_dict, keyed identically."""
    return {k: v.detach().to("cpu", dtype).clone().requires_grad_(requires_grad)
            for k, v in module.state_dict().items()}


def state_dict_shapes(input_dim=8, max_vertices=64, hidden_dims=(512, 1024, 2048, 1024),
                      output_dim=512, edge_hidden=512, vertex_dim=4, with_lazy=True):
    """name -> shape of the reference model's state_dict, derived
    from the constructor arithmetic; used to build parameters without
    constructing any nn.Module."""
    sd = {}

    def lin(name, o, i):
        sd[name + ".weight"] = (o, i)
        sd[name + ".bias"] = (o,)

    def ln(name, d):
        sd[name + ".weight"] = (d,)
        sd[name + ".bias"] = (d,)

    prev = input_dim
    for i, h in enumerate(hidden_dims):
        lin(f"encoder.mlp.{4 * i}", h, prev)
        ln(f"encoder.mlp.{4 * i + 1}", h)
        prev = h
    lin(f"encoder.mlp.{4 * len(hidden_dims)}", output_dim, prev)
    d = output_dim
    lin("encoder.feature_fusion.0", 4 * d, 2 * d); ln("encoder.feature_fusion.1", 4 * d)
    lin("encoder.feature_fusion.3", 2 * d, 4 * d); ln("encoder.feature_fusion.4", 2 * d)
    lin("encoder.feature_fusion.6", d, 2 * d)
    vp = "vertex_predictor."
    lin(vp + "vertex_mlp1.0", 4096, d); ln(vp + "vertex_mlp1.1", 4096)
    lin(vp + "vertex_mlp2.0", 2048, 4096); ln(vp + "vertex_mlp2.1", 2048)
    lin(vp + "vertex_mlp3.0", 2048, 2048); ln(vp + "vertex_mlp3.1", 2048)
    lin(vp + "vertex_mlp4.0", 1024, 2048); ln(vp + "vertex_mlp4.1", 1024)
    lin(vp + "final_layer", max_vertices * vertex_dim, 1024)
    lin(vp + "residual_proj1", 2048, d)
    lin(vp + "residual_proj2", 1024, d)
    if with_lazy:
        lin(vp + "point_pool_proj", d, 2 * d)
    ep = "edge_predictor."
    H = edge_hidden
    lin(ep + "vertex_proj.0", H // 2, 3); ln(ep + "vertex_proj.1", H // 2)
    lin(ep + "vertex_proj.3", H, H // 2); ln(ep + "vertex_proj.4", H)
    sd[ep + "attention.in_proj_weight"] = (3 * H, H)
    sd[ep + "attention.in_proj_bias"] = (3 * H,)
    lin(ep + "attention.out_proj", H, H)
    lin(ep + "spatial_proj.0", H // 4, 3)
    lin(ep + "spatial_proj.2", H // 4, H // 4)
    lin(ep + "edge_mlp.0", H, 2 * H + 7); ln(ep + "edge_mlp.1", H)
    lin(ep + "edge_mlp.4", H // 2, H); ln(ep + "edge_mlp.5", H // 2)
    lin(ep + "edge_mlp.8", H // 4, H // 2)
    lin(ep + "edge_mlp.10", 1, H // 4)
    return sd
