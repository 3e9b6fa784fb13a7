"""Diagnostic: one decision-frozen case (tests/test_frozen_grad_gpu.py::_run), per-tensor worst element and, for
edge_mlp.0.weight, the error by column block (F_i | F_j | c_i | c_j | distance).
    python scripts/debug_frozen_case.py bf16x3 1 33 3 3"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import helpers as H  # noqa: E402
from helpers import oracle  # noqa: E402
import test_frozen_grad_gpu as T  # noqa: E402

prec, B, N, V = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
counts = [int(c) for c in sys.argv[5:]]
from wf3d import config  # noqa: E402
from models.PointCloudToWireframe import PointCloudToWireframe  # noqa: E402

config.set_precision(prec)
config.SPLIT_MIN_ROWS = 1
dev = torch.device("cuda:0")
torch.manual_seed(20)
model = PointCloudToWireframe(8, V).to(dev).set_dropout(0.0)
model.train()
with torch.no_grad():
    for n, p in model.named_parameters():
        if p.dim() == 1:
            p.add_(0.05 * torch.randn(p.shape, generator=torch.Generator().manual_seed(len(n))).to(dev))
for seed in range(int(os.environ.get("SEED0", "100")), int(os.environ.get("SEED0", "100")) + 32):
    x, cnt, gen, _ = T._case(seed, B, N, V, counts)
    model.zero_grad(set_to_none=True)
    out = model(x.to(dev), cnt.to(dev))
    frozen, nb = H.capture_decisions(out, model)
    if nb == 0:
        break
cot = {k: torch.randn(out[k].shape, generator=gen) for k in ("vertices", "existence_probabilities", "edge_probs")}
sum((out[k] * cot[k].to(dev)).sum() for k in cot).backward()
P = oracle.params_from_module(model, dtype=torch.float64)
ref = oracle.model_forward(P, x.double(), cnt, V, training=True, frozen=frozen)
sum((ref[k] * cot[k].double()).sum() for k in cot).backward()
rows = []
for n, p in model.named_parameters():
    if P[n].grad is None:
        continue
    rows.append((H.elem_err(p.grad.cpu().numpy(), P[n].grad.numpy()), n))
for e, n in sorted(rows, reverse=True)[:8]:
    print(f"{e:.2e}  {n}")
g = model.edge_predictor.edge_mlp[0].weight.grad.cpu().double().numpy()
r = P["edge_predictor.edge_mlp.0.weight"].grad.numpy()
rms = np.sqrt((r * r).mean())
for name, sl in (("F_i", slice(0, 512)), ("F_j", slice(512, 1024)), ("c_i", slice(1024, 1027)), ("c_j", slice(1027, 1030)), ("dist", slice(1030, 1031))):
    d = np.abs(g[:, sl] - r[:, sl]) / np.maximum(np.abs(r[:, sl]), rms)
    print(f"  edge_mlp.0.weight[{name}]: worst {d.max():.2e}; rms of block {np.sqrt((r[:, sl] ** 2).mean()):.3e} (tensor rms {rms:.3e}); max |ref| {np.abs(r[:, sl]).max():.3e}")
