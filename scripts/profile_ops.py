"""Per-op GPU time of one cfg2 train step: every wf3d.ops wrapper is timed with its own pair of events
(synchronising after each call, so the numbers are kernel time without overlap), grouped by op and shapes."""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "wireframe-3d-prediction_amd"))
import torch  # noqa: E402
from wf3d import ops  # noqa: E402
import wf3d.functional as F  # noqa: E402
from models.PointCloudToWireframe import PointCloudToWireframe  # noqa: E402

dev = torch.device("cuda:0")
rec = collections.OrderedDict()
enabled = [False]


def wrap(name, fn):
    def inner(*a, **k):
        if not enabled[0]:
            return fn(*a, **k)
        shp = tuple(tuple(t.shape) for t in a if torch.is_tensor(t))[:3]
        extra = tuple(v for v in a if isinstance(v, int))[:1] + tuple((kk, vv) for kk, vv in k.items() if isinstance(vv, (int, bool)))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn(*a, **k)
        e1.record()
        torch.cuda.synchronize()
        key = (name, shp, extra)
        t = rec.setdefault(key, [0, 0.0])
        t[0] += 1
        t[1] += e0.elapsed_time(e1) * 1e3
        return out
    return inner


for n in dir(ops):
    f = getattr(ops, n)
    if callable(f) and not n.startswith("_") and getattr(f, "__module__", "") == ops.__name__ and n not in ("scratch", "check", "Pro", "EdgeMeta"):
        setattr(ops, n, wrap(n, f))

B, N, V = 32, 4096, 64
torch.manual_seed(0)
model = PointCloudToWireframe(input_dim=8, max_vertices=V).to(dev).train()
x = torch.randn(B, N, 8, device=dev)
counts = torch.full((B,), V, dtype=torch.int64)


def step():
    out = model(x, counts)
    loss = out["vertices"].sum() + out["existence_probabilities"].sum() + out["edge_probs"].sum()
    loss.backward()


for _ in range(2):
    step()
torch.cuda.synchronize()
enabled[0] = True
step()
enabled[0] = False
tot = sum(v[1] for v in rec.values())
print(f"total op time {tot / 1e3:.2f} ms over {sum(v[0] for v in rec.values())} calls")
thr = float(os.environ.get("MIN_US", 12))
for (name, shp, extra), (n, us) in sorted(rec.items(), key=lambda kv: -kv[1][1]):
    if us / n >= thr and max(s[0] for s in shp if s) < 100000:
        print(f"{us:8.1f} us  x{n}  {us / n:7.1f} each  {name} {shp} {extra}")
