"""Row f-2: wf3d.optim.ClipAdam (csrc/optim.hip) = clip_grad_norm_ + torch.optim.Adam of the reference's step tail
(train.py:96,141-142), including the reference's lazy-parameter quirk: a parameter that is in
model.parameters() but not in the optimizer enters the norm, gets its gradient scaled in place, and is never updated."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import helpers as H  # noqa: E402


def dev():
    return torch.device("cuda:0")


def _make(seed, shapes):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(*s, generator=g).to(dev()) for s in shapes]


@pytest.mark.parametrize("max_norm", [1.0, 1e9, None])
def test_clip_adam_matches_torch_over_several_steps(max_norm):
    from wf3d.optim import ClipAdam
    shapes = [(512, 8), (512,), (1024, 512), (7,), (3, 5, 11), (70001,), (2048, 1031)]
    pa = [torch.nn.Parameter(t.clone()) for t in _make(1, shapes)]
    pb = [torch.nn.Parameter(t.clone()) for t in _make(1, shapes)]
    lazy_a, lazy_b = torch.nn.Parameter(_make(2, [(512, 1024)])[0]), torch.nn.Parameter(_make(2, [(512, 1024)])[0])
    ref = torch.optim.Adam(pa, lr=1e-3, weight_decay=1e-6)
    opt = ClipAdam(pb, lr=1e-3, weight_decay=1e-6, max_norm=max_norm, norm_params=lambda: pb + [lazy_b])
    for step in range(4):
        grads = _make(10 + step, shapes + [(512, 1024)])
        scale = 3.0 if step % 2 else 0.01                     # both sides of the clip threshold
        ref.zero_grad(); opt.zero_grad()
        for p, q, g in zip(pa + [lazy_a], pb + [lazy_b], grads):
            # the lazy parameter's gradient ACCUMULATES (no optimizer ever zeroes it), as in the reference
            p.grad = g.clone() * scale if p.grad is None else p.grad + g * scale
            q.grad = g.clone() * scale if q.grad is None else q.grad + g * scale
        if max_norm is not None:
            n_ref = torch.nn.utils.clip_grad_norm_(pa + [lazy_a], max_norm=max_norm)
        ref.step()
        opt.step()
        if max_norm is not None:
            assert abs(float(opt.last_grad_norm) - float(n_ref)) <= 1e-5 * float(n_ref)
        for p, q in zip(pa + [lazy_a], pb + [lazy_b]):
            assert H.elem_err(q.detach().cpu().numpy(), p.detach().cpu().numpy()) < 2e-6
            assert H.elem_err(q.grad.cpu().numpy(), p.grad.cpu().numpy()) < 2e-6       # clipped in place, lazy one included
    for p, q in zip(pa, pb):
        sa, sb = ref.state[p], opt.state[q]
        assert int(sa["step"]) == int(sb["step"]) == 4
        assert H.elem_err(sb["exp_avg"].cpu().numpy(), sa["exp_avg"].cpu().numpy()) < 2e-6
        assert H.elem_err(sb["exp_avg_sq"].cpu().numpy(), sa["exp_avg_sq"].cpu().numpy()) < 2e-6
    assert torch.equal(lazy_b.detach(), _make(2, [(512, 1024)])[0])          # never updated
    # state_dict round trip into torch's Adam
    ref2 = torch.optim.Adam(pb, lr=1e-3, weight_decay=1e-6)
    ref2.load_state_dict(opt.state_dict())
    assert int(ref2.state[pb[0]]["step"]) == 4


def test_clip_adam_many_tensors_nan_norm_and_failed_step():
    """(a) 170 tensors: the launch is cut into several tensor tables (<= 80 each) that must share ONE global norm;
    (b) a NaN gradient poisons every gradient and parameter, as clip_grad_norm_ + Adam do;
    (c) a step that is refused (a clip-only gradient on the CPU) leaves every step counter where it was."""
    from wf3d.optim import ClipAdam
    shapes = [(37 + 3 * i,) if i % 3 else (5, 11 + i) for i in range(170)]
    pa = [torch.nn.Parameter(t.clone()) for t in _make(3, shapes)]
    pb = [torch.nn.Parameter(t.clone()) for t in _make(3, shapes)]
    ref = torch.optim.Adam(pa, lr=1e-3, weight_decay=1e-6)
    opt = ClipAdam(pb, lr=1e-3, weight_decay=1e-6, max_norm=1.0, norm_params=lambda: pb)
    for step in range(2):
        for p, q, g in zip(pa, pb, _make(20 + step, shapes)):
            p.grad, q.grad = g.clone(), g.clone()
        n_ref = torch.nn.utils.clip_grad_norm_(pa, max_norm=1.0)
        ref.step(); opt.step()
        assert abs(float(opt.last_grad_norm) - float(n_ref)) <= 1e-5 * float(n_ref)
        for p, q in zip(pa, pb):
            assert H.elem_err(q.detach().cpu().numpy(), p.detach().cpu().numpy()) < 2e-6
            assert H.elem_err(q.grad.cpu().numpy(), p.grad.cpu().numpy()) < 2e-6
    # (c) refused step: nothing advances
    bad = torch.nn.Parameter(torch.zeros(4))                       # a clip-only tensor whose gradient lives on the CPU
    bad.grad = torch.ones(4)
    opt2 = ClipAdam(pb, lr=1e-3, max_norm=1.0, norm_params=lambda: pb + [bad])
    opt2.load_state_dict(opt.state_dict())
    with pytest.raises(RuntimeError, match="fp32 CUDA gradients"):
        opt2.step()
    assert all(int(opt2.state[q]["step"]) == 2 for q in pb)
    # (b) NaN norm
    for p, q, g in zip(pa, pb, _make(30, shapes)):
        p.grad, q.grad = g.clone(), g.clone()
    pa[5].grad[0] = float("nan"); pb[5].grad[0] = float("nan")
    torch.nn.utils.clip_grad_norm_(pa, max_norm=1.0)
    ref.step(); opt.step()
    assert torch.isnan(opt.last_grad_norm)
    for p, q in zip(pa, pb):
        assert bool(torch.isnan(q.grad).all()) == bool(torch.isnan(p.grad).all()) is True
        assert bool(torch.isnan(q).all()) == bool(torch.isnan(p).all()) is True


def test_clip_adam_on_the_reference_trajectory():
    """tests/golden/traj.npz step 0 (reference model + reference loss + torch clip + torch Adam, generated from the imported
    reference): with ClipAdam in place of the two torch calls every tensor must move as far (update L1, 2e-3) and in the
    same direction (0.2 % of its elements) as the reference's, the lazy point_pool_proj must not move, and the reported
    pre-clip gradient norm must match."""
    from oracle import detgen
    from losses.WireframeLoss import WireframeLoss
    from models.PointCloudToWireframe import PointCloudToWireframe
    from wf3d.optim import ClipAdam
    g = H.load_golden("traj")
    seed, (B, N, V), counts = int(g["meta.seed"]), [int(v) for v in g["meta.dims"]], g["meta.counts"]
    x = detgen.normalish("traj.x", (B, N, 8), seed)
    x[1, ::5] = 0.0
    tv = 0.5 * detgen.normalish("traj.tv", (B, V, 3), seed)
    te = (np.arange(V)[None, :] < counts[:, None]).astype(np.float32)
    tl = (detgen.uniform("traj.tl", (B, V * (V - 1) // 2), 0, 1, seed) > 0.75).astype(np.float32)
    torch.manual_seed(seed)
    model = PointCloudToWireframe(input_dim=8, max_vertices=V).to(dev()).set_dropout(0.0)
    opt = ClipAdam(model.parameters(), lr=1e-3, weight_decay=1e-6, max_norm=1.0, norm_params=model.parameters)   # before the first forward
    crit = WireframeLoss(vertex_weight=3.0, edge_weight=1.5, existence_weight=1.0)
    xt, cnt = torch.from_numpy(x).to(dev()), torch.from_numpy(counts).to(dev())
    tgts = {"vertices": torch.from_numpy(tv).to(dev()), "vertex_existence": torch.from_numpy(te).to(dev()),
            "edge_labels": torch.from_numpy(tl).to(dev()), "vertex_counts": cnt}
    model.train()
    opt.zero_grad()
    crit(model(xt, cnt), tgts)["total_loss"].backward()
    names = [k for k, _ in model.named_parameters()]
    before = [p.detach().clone() for p in model.parameters()]
    opt.step()
    assert abs(float(opt.last_grad_norm) - float(g["traj"][0][4])) <= 2e-3 * float(g["traj"][0][4])
    for n, p, b, l1, up in zip(names, model.parameters(), before, g["step0.update_l1"], g["step0.moved_up"]):
        d = p.detach() - b
        assert abs(float(d.double().abs().sum()) - l1) <= 2e-3 * max(l1, 1e-9), (n, l1)
        if n != "edge_predictor.attention.in_proj_bias":
            assert abs(int((d > 0).sum()) - int(up)) <= max(2, 0.002 * p.numel()), (n, int((d > 0).sum()), int(up))
    lazy = model.vertex_predictor.point_pool_proj
    assert float((lazy.weight.detach() - before[names.index("vertex_predictor.point_pool_proj.weight")]).abs().max()) == 0.0
