"""Which parameters does the first train step treat differently from the reference fixture (tests/golden/traj.npz)?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import helpers as H
from oracle import detgen
from losses.WireframeLoss import WireframeLoss
from models.PointCloudToWireframe import PointCloudToWireframe
g = np.load(os.path.join(ROOT, "tests/golden/traj.npz"))
seed, (B, N, V), counts = int(g["meta.seed"]), [int(v) for v in g["meta.dims"]], g["meta.counts"]
dev = torch.device("cuda:0")
x = detgen.normalish("traj.x", (B, N, 8), seed); x[1, ::5] = 0.0
tv = 0.5 * detgen.normalish("traj.tv", (B, V, 3), seed)
te = (np.arange(V)[None, :] < counts[:, None]).astype(np.float32)
tl = (detgen.uniform("traj.tl", (B, V * (V - 1) // 2), 0, 1, seed) > 0.75).astype(np.float32)
torch.manual_seed(seed)
model = PointCloudToWireframe(input_dim=8, max_vertices=V).to(dev).set_dropout(0.0)
opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-6)
crit = WireframeLoss(vertex_weight=3.0, edge_weight=1.5, existence_weight=1.0)
xt, cnt = torch.from_numpy(x).to(dev), torch.from_numpy(counts).to(dev)
tgts = {"vertices": torch.from_numpy(tv).to(dev), "vertex_existence": torch.from_numpy(te).to(dev),
        "edge_labels": torch.from_numpy(tl).to(dev), "vertex_counts": cnt}
model.train()
opt.zero_grad()
out = model(xt, cnt)
crit(out, tgts)["total_loss"].backward()
names = [k for k, _ in model.named_parameters()]
assert names == list(g["step0.names"]), (len(names), len(g["step0.names"]))
g0 = np.array([float(p.grad.double().norm()) if p.grad is not None else -1.0 for p in model.parameters()])
torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
before = [p.detach().clone() for p in model.parameters()]
opt.step()
d0 = np.array([float((p.detach() - b).double().abs().sum()) for p, b in zip(model.parameters(), before)])
up0 = np.array([int(((p.detach() - b) > 0).sum()) for p, b in zip(model.parameters(), before)])
tot = 0
for n, u, ur, p in zip(names, up0, g["step0.moved_up"], model.parameters()):
    tot += abs(int(u) - int(ur))
    if abs(int(u) - int(ur)) > 0.002 * p.numel(): print(f"{n}: moved-up count {u} vs reference {ur} of {p.numel()}")
print("sum over tensors of |moved-up count difference|:", tot, "of", sum(p.numel() for p in model.parameters()))
for n, a, b, da, db, p in zip(names, g0, g["step0.grad_norm"], d0, g["step0.update_l1"], model.parameters()):
    flag = ""
    if (a < 0) != (b < 0): flag += " NONE-MISMATCH"
    if b > 0 and abs(a - b) > 1e-3 * b: flag += f" grad {a:.6g} vs {b:.6g}"
    if abs(da - db) > 2e-3 * max(db, 1e-9): flag += f" update_l1 {da:.6g} vs {db:.6g} (numel {p.numel()})"
    if flag: print(n, flag)
print("done")
