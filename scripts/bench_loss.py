"""Row f-1: GPU time of the device WireframeLoss pieces at the cfg2 shape (B=32, V=64, E=2016)."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wireframe-3d-prediction_amd"))
import torch
from wf3d import ops
from losses.WireframeLoss import WireframeLoss
dev = torch.device("cuda:0")
B, V = 32, int(os.environ.get("V", 64))
E = V * (V - 1) // 2
torch.manual_seed(0)
pv = torch.randn(B, V, 4, device=dev)[:, :, :3].requires_grad_()
pe = torch.rand(B, V, device=dev).requires_grad_()
pp = torch.rand(B, E, device=dev).requires_grad_()
counts = torch.randint(4, V + 1, (B,), device=dev)
tgts = {"vertices": torch.randn(B, V, 3, device=dev), "vertex_existence": (torch.arange(V, device=dev)[None] < counts[:, None]).float(),
        "edge_labels": (torch.rand(B, E, device=dev) > 0.8).float(), "vertex_counts": counts}


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return statistics.median(ts)


pec = pe.detach().contiguous()
cost = ops.loss_cost_matrix(pv.detach(), pec, tgts["vertices"], counts)
print(f"cost matrix      {timeit(lambda: ops.loss_cost_matrix(pv.detach(), pec, tgts['vertices'], counts)):8.1f} us")
print(f"assignment (JV)  {timeit(lambda: ops.loss_assign(cost)):8.1f} us")
c4r = ops.loss_assign(cost)
print(f"terms + grads    {timeit(lambda: ops.loss_terms_assigned(pv.detach(), pec, pp.detach(), tgts['vertices'], tgts['vertex_existence'], tgts['edge_labels'], c4r, counts, (3.0, 1.0, 1.5))):8.1f} us")
for mode in ("device", "scipy"):
    crit = WireframeLoss(3.0, 1.5, 1.0, assignment=mode)
    def full():
        for t in (pv, pe, pp):
            t.grad = None
        crit({"vertices": pv, "existence_probabilities": pe, "edge_probs": pp}, tgts)["total_loss"].backward()
    print(f"WireframeLoss fwd+bwd, assignment={mode}: {timeit(full):8.1f} us")
