"""Row f-1 (WireframeLoss): the CPU oracle against the reference-generated fixtures (CPU), and the
device implementation against the same fixtures (GPU)."""
import numpy as np
import pytest
import torch

import helpers as H
from helpers import detgen
from oracle import loss_cpu

CASES = ["loss_a", "loss_b", "loss_c"]


def build(tag, device="cpu"):
    g = H.load_golden(tag)
    B, V, seed = int(g["meta.B"]), int(g["meta.V"]), int(g["meta.seed"])
    ep, et = int(g["meta.max_e_pred"]), int(g["meta.max_e_tgt"])
    pv = torch.from_numpy(detgen.normalish(tag + ".pv", (B, V, 3), seed)).to(device).requires_grad_()
    pe = torch.sigmoid(torch.from_numpy(2.0 * detgen.normalish(tag + ".pe", (B, V), seed))).to(device).requires_grad_()
    pp = torch.sigmoid(torch.from_numpy(2.0 * detgen.normalish(tag + ".pp", (B, ep), seed))).to(device).requires_grad_()
    tv = torch.from_numpy(detgen.normalish(tag + ".tv", (B, V, 3), seed)).to(device)
    cnt = torch.tensor(g["meta.counts"].tolist(), dtype=torch.long, device=device)
    te = (torch.arange(V, device=device)[None, :] < cnt[:, None]).float()
    tl = (torch.from_numpy(detgen.uniform(tag + ".tl", (B, et), 0, 1, seed)) > 0.7).float().to(device)
    preds = {"vertices": pv, "existence_probabilities": pe, "edge_probs": pp}
    tgts = {"vertices": tv, "vertex_existence": te, "edge_labels": tl, "vertex_counts": cnt}
    return g, preds, tgts


def check(g, out, matches, preds, tol):
    for k in ("total_loss", "vertex_loss", "existence_loss", "edge_loss"):
        assert abs(float(out[k]) - float(g["out." + k])) < tol * max(1.0, abs(float(g["out." + k]))), k
    assert [len(m[0]) for m in matches] == g["match.lens"].tolist()
    assert np.array_equal(np.concatenate([np.asarray(m[0]) for m in matches]), g["match.pred"])     # bit-exact assignment
    assert np.array_equal(np.concatenate([np.asarray(m[1]) for m in matches]), g["match.tgt"])
    out["total_loss"].backward()
    for name, key in (("vertices", "grad.vertices"), ("existence_probabilities", "grad.existence"), ("edge_probs", "grad.edge_probs")):
        assert H.rel_err(preds[name].grad.cpu().numpy(), g[key]) < 10 * tol, name


@pytest.mark.parametrize("tag", CASES)
def test_loss_oracle_matches_reference_fixture(tag):
    g, preds, tgts = build(tag)
    wv, wd, we = g["meta.weights"].tolist()
    out, matches = loss_cpu.wireframe_loss(preds, tgts, vertex_weight=wv, edge_weight=wd, existence_weight=we)
    check(g, out, matches, preds, 1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("V,B", [(2, 3), (7, 5), (26, 4), (64, 32), (200, 3), (256, 2)])
def test_device_assignment_is_optimal_like_scipy(V, B):
    """Device Jonker-Volgenant vs scipy on wireframe-shaped cost matrices (real + identical dummy columns)."""
    from scipy.optimize import linear_sum_assignment
    from wf3d import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(V * 31 + B)
    pv, tv = torch.randn(B, V, 3, generator=g).to(dev), torch.randn(B, V, 3, generator=g).to(dev)
    pe = torch.rand(B, V, generator=g).to(dev)
    counts = torch.randint(0, V + 1, (B,), generator=g).to(dev)
    counts[0] = V
    cost = ops.loss_cost_matrix(pv, pe, tv, counts)
    cn, cnt = cost.cpu().numpy().astype(np.float64), counts.cpu().tolist()
    # square problem (what the loss uses): same matches as scipy, exact ties included (same algorithm, same row order);
    # rectangular shortcut over the real targets only: an optimum too (equal cost), but L1 costs have exact ties
    # — 2 of 32 samples at V = 64 — which it may resolve differently
    for rect, c4r in ((False, ops.loss_assign(cost).cpu().numpy()), (True, ops.loss_assign(cost, counts).cpu().numpy())):
        for b in range(B):
            assert sorted(c4r[b].tolist()) == list(range(V))                      # a permutation
            ri, ci = linear_sum_assignment(cn[b])
            ours = cn[b][np.arange(V), c4r[b]].sum()
            assert abs(ours - cn[b][ri, ci].sum()) <= 1e-12 * max(1.0, abs(ours))    # same optimum
            if not rect:
                real_ref = {(int(r), int(c)) for r, c in zip(ri, ci) if c < cnt[b]}
                real_our = {(p, int(c4r[b][p])) for p in range(V) if c4r[b][p] < cnt[b]}
                assert real_ref == real_our                                       # same matches to real targets


@pytest.mark.gpu
@pytest.mark.parametrize("assignment", ["device", "scipy"])
@pytest.mark.parametrize("tag", CASES)
def test_device_loss_matches_reference_fixture(tag, assignment):
    from losses.WireframeLoss import WireframeLoss
    g, preds, tgts = build(tag, "cuda:0")
    wv, wd, we = g["meta.weights"].tolist()
    crit = WireframeLoss(vertex_weight=wv, edge_weight=wd, existence_weight=we, assignment=assignment)
    out = crit(preds, tgts)
    assert set(out) == {"total_loss", "vertex_loss", "existence_loss", "edge_loss"}
    matches = crit._hungarian_matching(preds, tgts)
    check(g, out, matches, preds, 2e-6)


@pytest.mark.gpu
def test_device_loss_drives_the_model_backward():
    """total_loss.backward() through the drop-in model, strided `vertices` view included."""
    from losses.WireframeLoss import WireframeLoss
    from models.PointCloudToWireframe import PointCloudToWireframe
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    B, N, V = 3, 256, 12
    model = PointCloudToWireframe(8, V).to(dev).set_dropout(0.0)
    model.train()
    x = torch.randn(B, N, 8, device=dev)
    cnt = torch.tensor([12, 3, 7], device=dev)
    out = model(x, cnt)
    tg = {"vertices": torch.randn(B, V, 3, device=dev), "vertex_existence": (torch.arange(V, device=dev)[None] < cnt[:, None]).float(),
          "edge_labels": (torch.rand(B, V * (V - 1) // 2, device=dev) > 0.8).float(), "vertex_counts": cnt}
    crit = WireframeLoss(3.0, 1.5, 1.0)
    ld = crit(out, tg)
    ld["total_loss"].backward()
    # same numbers from the CPU oracle loss on the same predictions
    preds_cpu = {k: out[k].detach().cpu().requires_grad_() for k in ("vertices", "existence_probabilities", "edge_probs")}
    ref, _ = loss_cpu.wireframe_loss(preds_cpu, {k: v.cpu() for k, v in tg.items()}, 3.0, 1.5, 1.0)
    assert abs(float(ld["total_loss"]) - float(ref["total_loss"])) < 1e-5 * float(ref["total_loss"])
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for n, p in model.named_parameters() if "spatial_proj" not in n)


@pytest.mark.gpu
def test_train_meter_tracks_the_reference_loop_quantities_without_syncs():
    """wf3d.meter.TrainMeter against train.py:145-157 evaluated on the host: loss history, best loss, the monitoring RMSE of
    sample 0 over its first counts[0] vertices and its running minimum, the last loss terms — read back in one copy."""
    import numpy as np
    from wf3d.meter import TrainMeter
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    B, V = 3, 9
    counts = torch.tensor([7, 2, 9])
    meter = TrainMeter(dev, history=4)
    target = torch.randn(B, V, 3, generator=g)
    hist, best_loss, best_rmse = [], float("inf"), float("inf")
    for step in range(7):
        out4 = torch.randn(B, V, 4, generator=g).to(dev)                # `vertices` is a strided view of [B, V, 4] in the model
        losses = {k: torch.rand((), generator=g).to(dev) * (3 - 0.3 * step) for k in ("total_loss", "vertex_loss", "existence_loss", "edge_loss")}
        meter.update(losses, out4[:, :, :3], target.to(dev), counts.to(dev))
        pv = out4[0, :7, :3].cpu().numpy()
        rmse = float(np.sqrt(np.mean((pv - target[0, :7].numpy()) ** 2)))
        hist.append(float(losses["total_loss"]))
        best_loss, best_rmse = min(best_loss, hist[-1]), min(best_rmse, rmse)
    m = meter.read()
    assert m["steps"] == 7
    assert abs(m["best_loss"] - best_loss) < 1e-6 and abs(m["best_vertex_rmse"] - best_rmse) < 1e-6
    assert abs(m["vertex_rmse"] - rmse) < 1e-6 and abs(m["total_loss"] - hist[-1]) < 1e-7
    assert abs(m["edge_loss"] - float(losses["edge_loss"])) < 1e-7
    assert np.allclose(m["loss_history"], hist[-4:], atol=1e-7)          # ring of the last `history` steps, oldest first


@pytest.mark.gpu
@pytest.mark.parametrize("B,V,counts,ep,et", [(1, 1, [1], 1, 1), (1, 2, [0], 1, 1), (2, 3, [0, 0], 3, 3), (5, 4, [4, 1, 0, 2, 3], 6, 4),
                                              (64, 3, None, 3, 3), (3, 130, [130, 1, 77], 400, 8385), (2, 256, [256, 200], 32640, 32640),
                                              (7, 17, None, 136, 50)])
def test_device_loss_matches_the_cpu_oracle_at_other_shapes(B, V, counts, ep, et):
    """Losses, assignment and input gradients against oracle/loss_cpu.py (pinned by the reference's fixtures above) at shapes the
    fixtures do not hold: single samples and vertices, batches without any target vertex, 64 samples, 130 / 256 vertices, edge
    widths that differ between prediction and target in either direction."""
    from losses.WireframeLoss import WireframeLoss
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(B * 1000 + V)
    if counts is None:
        counts = torch.randint(0, V + 1, (B,), generator=g).tolist()
    cnt = torch.tensor(counts, dtype=torch.long)
    pv, tv = torch.randn(B, V, 3, generator=g), torch.randn(B, V, 3, generator=g)
    pe = torch.sigmoid(2.0 * torch.randn(B, V, generator=g))
    pp = torch.sigmoid(2.0 * torch.randn(B, ep, generator=g))
    tl = (torch.rand(B, et, generator=g) > 0.7).float()
    te = (torch.arange(V)[None, :] < cnt[:, None]).float()
    tg = {"vertices": tv, "vertex_existence": te, "edge_labels": tl, "vertex_counts": cnt}
    pc = {"vertices": pv.clone().requires_grad_(), "existence_probabilities": pe.clone().requires_grad_(), "edge_probs": pp.clone().requires_grad_()}
    ref, ref_m = loss_cpu.wireframe_loss(pc, tg, 3.0, 1.5, 1.0)
    ref["total_loss"].backward()
    pd = {k: v.detach().clone().to(dev).requires_grad_() for k, v in pc.items()}
    crit = WireframeLoss(3.0, 1.5, 1.0)
    out = crit(pd, {k: v.to(dev) for k, v in tg.items()})
    for k in ("total_loss", "vertex_loss", "existence_loss", "edge_loss"):
        assert abs(float(out[k]) - float(ref[k])) < 5e-6 * max(1.0, abs(float(ref[k]))), (k, float(out[k]), float(ref[k]))
    out["total_loss"].backward()
    for k in pc:
        a, b = pd[k].grad, pc[k].grad
        if b is None:                                   # e.g. no target vertex anywhere: the vertex term does not depend on the coordinates
            assert a is None or float(a.abs().max()) == 0.0, k
            continue
        assert a is not None, k
        assert float((a.cpu() - b).abs().max()) <= 2e-5 * max(float(b.abs().max()), 1e-12), k
