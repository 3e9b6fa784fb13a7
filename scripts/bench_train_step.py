"""Row f-1 measurement: a step WITH the loss (cfg2 shape).  Compares
  (a) the metric's step: model fwd + bwd from a fixed cotangent,
  (b) fwd + device WireframeLoss (losses/WireframeLoss.py here) + bwd,
  (c) fwd + the reference formulation of the loss in torch ops on the GPU (oracle/loss_cpu.py run
      on CUDA tensors: per-sample cdist / cat / .cpu() sync / scipy, as WireframeLoss.py:130-236) + bwd.
"""
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "wireframe-3d-prediction_amd"))
import torch  # noqa: E402
from oracle import loss_cpu  # noqa: E402
from losses.WireframeLoss import WireframeLoss  # noqa: E402
from models.PointCloudToWireframe import PointCloudToWireframe  # noqa: E402

dev = torch.device("cuda:0")
B, N, V = 32, 4096, 64
torch.manual_seed(1234)
model = PointCloudToWireframe(8, V).to(dev)
model.train()
g = torch.Generator().manual_seed(1)
x = torch.randn(B, N, 8, generator=g).to(dev)
counts = torch.randint(4, V + 1, (B,), generator=g)
counts[0] = V
counts = counts.to(dev)
E = V * (V - 1) // 2
tg = {"vertices": torch.randn(B, V, 3, generator=g).to(dev),
      "vertex_existence": (torch.arange(V)[None] < counts.cpu()[:, None]).float().to(dev),
      "edge_labels": (torch.rand(B, E, generator=g) > 0.8).float().to(dev), "vertex_counts": counts}
crit = WireframeLoss(3.0, 1.5, 1.0)
cot = None


def step_a():
    global cot
    out = model(x, counts)
    if cot is None:
        cot = {k: torch.randn_like(out[k]) for k in ("vertices", "existence_probabilities", "edge_probs")}
    sum((out[k] * cot[k]).sum() for k in cot).backward()


def step_b():
    crit(model(x, counts), tg)["total_loss"].backward()


def step_c():
    out = model(x, counts)
    ld, _ = loss_cpu.wireframe_loss(out, tg, 3.0, 1.5, 1.0)
    ld["total_loss"].backward()


for name, fn in (("a: fwd+bwd (metric)", step_a), ("b: fwd + device loss + bwd", step_b), ("c: fwd + reference-style torch loss + bwd", step_c)):
    for _ in range(3):
        model.zero_grad(set_to_none=True)
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        model.zero_grad(set_to_none=True)
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    print(f"{name:45s} median {statistics.median(ts):7.2f} ms/step   {B / statistics.median(ts) * 1e3:7.1f} clouds/s")
