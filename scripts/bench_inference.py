"""Forward-only (eval, no_grad) throughput of the drop-in model at the cfg2 shape — the evaluate.py:71 call."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wireframe-3d-prediction_amd"))
import torch
from models.PointCloudToWireframe import PointCloudToWireframe
dev = torch.device("cuda:0")
B, N, V = 32, 4096, 64
torch.manual_seed(0)
model = PointCloudToWireframe(input_dim=8, max_vertices=V).to(dev)
x = torch.randn(B, N, 8, device=dev)
counts = torch.full((B,), V, dtype=torch.int64)
for mode in ("train-mode forward (counts given)", "eval forward (data-dependent counts)"):
    if mode.startswith("eval"):
        model.eval()
        with torch.no_grad():                      # make the untrained model predict >= 2 vertices per sample
            model.vertex_predictor.final_layer.bias.view(V, 4)[:, 3].fill_(3.0)
        fn = lambda: model(x)
    else:
        model.train().set_dropout(0.0)
        fn = lambda: model(x, counts)
    with torch.no_grad():
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(10):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
    t = statistics.median(ts)
    print(f"{mode}: {t:.2f} ms / batch of {B} = {B / t * 1e3:.0f} clouds/s")
