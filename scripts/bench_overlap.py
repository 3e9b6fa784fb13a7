"""Can the HBM-bound ln_prep of one row chunk run UNDER the MFMA-bound GEMM of the other chunk (two streams)?
Chain: X[M,1024] -> GEMM(2048) -> ln_prep -> GEMM(1024) -> ln_prep -> GEMM(512), sequential vs two-stream pipelined."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wireframe-3d-prediction_amd"))
import torch
from wf3d import ops
dev = torch.device("cuda:0")
M = 131072
torch.manual_seed(0)
dims = [1024, 2048, 1024, 512]
X = ops.split_rows(torch.randn(M, dims[0], device=dev))
Ws = [ops.split_rows(torch.randn(dims[i + 1], dims[i], device=dev) * 0.03) for i in range(3)]
gs = [torch.ones(d, device=dev) for d in dims[1:]]
bs = [torch.zeros(d, device=dev) for d in dims[1:]]
zs = [torch.empty(M, d, device=dev) for d in dims[1:]]
side = torch.cuda.Stream(dev)


def sequential(chunks):
    rows = M // chunks
    for c in range(chunks):
        sl = slice(c * rows, (c + 1) * rows)
        a = X[sl]
        for l in range(3):
            ops.gemm_split(a, Ws[l], out=zs[l][sl])
            if l < 2:
                _, _, a = ops.ln_prep(zs[l][sl], gs[l], bs[l], ops.ACT_RELU)


def pipelined(chunks=2):
    """main stream: all GEMMs in (layer, chunk) order; side stream: all ln_preps; events carry the dependencies."""
    main = torch.cuda.current_stream(dev)
    rows = M // chunks
    keep = []
    a = [X[c * rows:(c + 1) * rows] for c in range(chunks)]
    for l in range(3):
        nxt = [None] * chunks
        for c in range(chunks):
            sl = slice(c * rows, (c + 1) * rows)
            if l > 0:
                main.wait_event(a[c][1])             # ln_prep(l-1, c) done
                ops.gemm_split(a[c][0], Ws[l], out=zs[l][sl])
            else:
                ops.gemm_split(a[c], Ws[l], out=zs[l][sl])
            if l < 2:
                ev = torch.cuda.Event(); ev.record(main)
                side.wait_event(ev)
                with torch.cuda.stream(side):
                    _, _, h = ops.ln_prep(zs[l][sl], gs[l], bs[l], ops.ACT_RELU)
                    done = torch.cuda.Event(); done.record(side)
                keep.append(h)
                nxt[c] = (h, done)
        a = nxt
    main.wait_stream(side)
    return keep


def timeit(fn, n=7):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
        del r
    return statistics.median(ts)


for name, fn in [("sequential, 1 chunk", lambda: sequential(1)), ("sequential, 2 chunks", lambda: sequential(2)),
                 ("two streams, 2 chunks", lambda: pipelined(2)), ("two streams, 4 chunks", lambda: pipelined(4)),
                 ("sequential, 1 chunk", lambda: sequential(1))]:
    print(f"{name:24s} {timeit(fn):7.3f} ms")
