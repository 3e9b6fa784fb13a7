"""Drop-in behaviour under the reference's training sequence (train.py:40,96-142): model built,
Adam created BEFORE the first forward (so the lazy point_pool_proj is never optimised),
zero_grad / forward / backward / clip_grad_norm_ / step.  The loss here is a plain differentiable
stand-in (the reference's Hungarian loss is host-side scipy, out of scope)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

import helpers as H  # noqa: E402,F401


def test_reference_training_sequence_runs_and_overfits():
    from models.PointCloudToWireframe import PointCloudToWireframe
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    Bt, Nt, V = 3, 2560, 26                            # main.py: batch 3, 2560 points; a real first batch had V=26
    model = PointCloudToWireframe(input_dim=8, max_vertices=V).to(dev)
    n_before = sum(p.numel() for p in model.parameters())
    assert n_before == 30373097                        # what train.py:43-45 logs for V=26
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-6)
    model.train()
    x = torch.randn(Bt, Nt, 8, device=dev)
    counts = torch.tensor([26, 4, 24], device=dev)
    tgt_v = torch.randn(Bt, V, 3, device=dev)
    tgt_e = (torch.rand(Bt, V * (V - 1) // 2, device=dev) > 0.8).float()
    losses = []
    for step in range(12):
        opt.zero_grad()
        out = model(x, counts)
        e = out["edge_probs"]
        loss = torch.nn.functional.smooth_l1_loss(out["vertices"], tgt_v) \
            + torch.nn.functional.binary_cross_entropy(e, tgt_e[:, :e.shape[1]]) \
            + torch.nn.functional.binary_cross_entropy(out["existence_probabilities"], torch.ones(Bt, V, device=dev))
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.0)
        opt.step()
        losses.append(loss.item())
    assert all(torch.isfinite(torch.tensor(losses)))
    assert losses[-1] < losses[0]
    lazy = model.vertex_predictor.point_pool_proj
    opt_ids = {id(p) for g in opt.param_groups for p in g["params"]}
    assert id(lazy.weight) not in opt_ids               # created after Adam: never optimised, like the reference
    assert lazy.weight.grad is not None                  # ... but its gradient accumulates (never zeroed)
    assert sum(p.numel() for p in model.parameters()) == 30897897
    sd = model.state_dict()
    assert "vertex_predictor.point_pool_proj.weight" in sd and len(sd) == 80
    # evaluate.py:49-56 reload path: strict=False into a fresh model, eval forward
    fresh = PointCloudToWireframe(8, sd["vertex_predictor.final_layer.weight"].shape[0] // 4).to(dev)
    missing = fresh.load_state_dict(sd, strict=False)
    assert set(missing.unexpected_keys) == {"vertex_predictor.point_pool_proj.weight", "vertex_predictor.point_pool_proj.bias"}
    fresh.eval()
    with torch.no_grad():
        try:
            o = fresh(x)
            assert o["edge_probs"].shape[0] == Bt
        except IndexError:
            pass                                        # data-dependent count <= 1 crashes the reference too (§9 Q4)
