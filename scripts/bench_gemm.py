"""Micro-benchmark of wf3d_gemm at the encoder's shapes (cfg2: M = 32*4096 rows).
Interleaved rounds in one process, random data, median per shape (TFLOP/s)."""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wireframe-3d-prediction_amd"))
import torch  # noqa: E402
from wf3d import ops  # noqa: E402

dev = torch.device("cuda:0")
M = int(os.environ.get("M", 131072))
layers = [(512, 1024), (1024, 2048), (2048, 1024), (1024, 512)]   # (K_in, N_out)
torch.manual_seed(0)
cases = []
for K, N in layers:
    X = torch.randn(M, K, device=dev)
    W = torch.randn(N, K, device=dev) * 0.05
    G = torch.randn(M, N, device=dev)
    mu, rs = ops.row_stats(X)
    gam, bet = torch.ones(K, device=dev), torch.zeros(K, device=dev)
    pro = ops.Pro(ops.ACT_RELU, mu, rs, gam, bet)
    o_nt, o_nn, o_tn = torch.empty(M, N, device=dev), torch.empty(M, K, device=dev), torch.empty(N, K, device=dev)
    fl = 2.0 * M * N * K
    cases.append((f"NT+pro K={K} N={N}", lambda X=X, W=W, pro=pro, o=o_nt: ops.gemm(X, W, ops.NT, pro=pro, out=o), fl))
    cases.append((f"NT     K={K} N={N}", lambda X=X, W=W, o=o_nt: ops.gemm(X, W, ops.NT, out=o), fl))
    cases.append((f"NN     K={N} N={K}", lambda G=G, W=W, o=o_nn: ops.gemm(G, W, ops.NN, out=o), fl))
    Xs, Ws, Gs, WTs = ops.split_rows(X), ops.split_rows(W), ops.split_rows(G), ops.split_rows(W, transpose=True)
    bias = torch.randn(N, device=dev)
    cases.append((f"SPLIT NT K={K} N={N}", lambda Xs=Xs, Ws=Ws, o=o_nt: ops.gemm_split(Xs, Ws, out=o), fl))
    cases.append((f"SPLIT NT+bias K={K} N={N}", lambda Xs=Xs, Ws=Ws, o=o_nt, b=bias: ops.gemm_split(Xs, Ws, bias=b, out=o), fl))
    cases.append((f"SPLIT dgrad K={N} N={K}", lambda Gs=Gs, WTs=WTs, o=o_nn: ops.gemm_split(Gs, WTs, out=o), fl))
    Gts, Xts = ops.split_transpose(Gs, in_sx8=True), ops.split_transpose(Xs, in_sx8=True)
    cases.append((f"SPLIT wgrad(NT on transposes) {N}x{K}", lambda a=Gts, b=Xts, o=o_tn: ops.gemm_split(a, b, out=o), fl))
    cases.append((f"SPLIT wgrad(TN tr-read) {N}x{K}", lambda a=Gs, b=Xs, o=o_tn: ops.gemm_split_tn(a, b, out=o), fl))
    cases.append((f"SPLIT transposes for wgrad {N}x{K}", lambda a=Gs, b=Xs: (ops.split_transpose(a, in_sx8=True), ops.split_transpose(b, in_sx8=True)), fl))
    cases.append((f"TN+pro K={M} out={N}x{K}", lambda G=G, X=X, pro=pro, o=o_tn: ops.gemm(G, X, ops.TN, pro=pro, out=o), fl))
if os.environ.get("ONLY"):
    cases = [c for c in cases if os.environ["ONLY"] in c[0]]
times = {c[0]: [] for c in cases}
for c in cases:
    c[1]()
torch.cuda.synchronize()
for r in range(int(os.environ.get("ROUNDS", 5))):
    for name, fn, fl in cases:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        times[name].append(e0.elapsed_time(e1))
tot = 0.0
for name, fn, fl in cases:
    ms = statistics.median(times[name])
    if not name.startswith("NT  ") and not name.startswith("SPLIT"):
        tot += ms
    print(f"{name:34s} {ms:8.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s")
print(f"sum (NT+pro, NN, TN+pro) = {tot:.2f} ms")
