"""Helper run in a fresh process by tests/test_variants_gpu.py: with WF3D_SPLIT_DMA / WF3D_TN16 set in the
environment (they are read once per process), checks the selected split-GEMM kernel against fp64."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import helpers as H  # noqa: E402,F401  (sets sys.path for the package)
import torch  # noqa: E402
from wf3d import ops  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
worst = 0.0
for M, N, K in [(256, 256, 32), (512, 384, 96), (1000, 200, 64), (2048, 512, 512)]:
    A, B = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
    bias = torch.randn(N, device=dev)
    got = ops.gemm_split(ops.split_rows(A), ops.split_rows(B), bias=bias)
    want = A.double() @ B.double().T + bias.double()
    worst = max(worst, float((got.double() - want).abs().max() / want.abs().max()))
    Ai = (torch.arange(M * K, device=dev).reshape(M, K) % 9 - 4).float()
    Bi = (torch.arange(N * K, device=dev).reshape(N, K) % 5 - 2).float()
    gi = ops.gemm_split(ops.split_rows(Ai), ops.split_rows(Bi))
    assert torch.equal(gi.double(), Ai.double() @ Bi.double().T), ("integer data not exact", M, N, K)
for K, Mo, No in [(64, 256, 256), (4096, 512, 256), (96, 256, 128)]:
    A, B = torch.randn(K, Mo, device=dev), torch.randn(K, No, device=dev)
    got = ops.gemm_split_tn(ops.split_rows(A), ops.split_rows(B))
    want = A.double().T @ B.double()
    worst = max(worst, float((got.double() - want).abs().max() / want.abs().max()))
assert worst < 3e-5, worst
print("OK", worst)
