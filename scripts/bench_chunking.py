"""Does row-chunking the per-point MLP keep producer->consumer tensors in the 256 MB Infinity Cache?
Times layer l -> ln_prep -> layer l+1 of the encoder (1024 -> 2048 -> 1024) over all B*N rows, either
in one pass or in row chunks through both layers."""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wireframe-3d-prediction_amd"))
import torch  # noqa: E402
from wf3d import ops  # noqa: E402

dev = torch.device("cuda:0")
M = 131072
torch.manual_seed(0)
X = torch.randn(M, 1024, device=dev)
W1 = torch.randn(2048, 1024, device=dev) * 0.03
W2 = torch.randn(1024, 2048, device=dev) * 0.03
g, b = torch.ones(2048, device=dev), torch.zeros(2048, device=dev)
Xs, W1s, W2s = ops.split_rows(X), ops.split_rows(W1), ops.split_rows(W2)
z1 = torch.empty(M, 2048, device=dev)
z2 = torch.empty(M, 1024, device=dev)


def run(chunks):
    rows = M // chunks
    for c in range(chunks):
        sl = slice(c * rows, (c + 1) * rows)
        ops.gemm_split(Xs[sl], W1s, out=z1[sl])
        mu, rs, h = ops.ln_prep(z1[sl], g, b, ops.ACT_RELU)
        ops.gemm_split(h, W2s, out=z2[sl])


for chunks in (1, 2, 4, 8, 16, 1):
    run(chunks)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        run(chunks)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print(f"chunks={chunks:3d} rows/chunk={M // chunks:7d}  {statistics.median(ts):8.3f} ms")
