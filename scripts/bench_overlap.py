#!/usr/bin/env python3
"""Does a bandwidth-bound row pass hide beside the persistent split GEMM?  The GEMM (131072 x 1024 x 1024, bias) is
launched on `g` CUs (wf3d_set_option("gemm_cus", g)) on one stream, ln_prep / ln_act_bwd over a 131072 x 1024 block
on a second stream right after it; both are timed alone and together."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wireframe-3d-prediction_amd"))
from wf3d import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
M, K, N, D = 131072, 1024, 1024, 1024
side = torch.cuda.Stream(dev)
lib = _lib.load()


def run(gemm_reps, row_reps, gemm, row):
    """elapsed (us) of: GEMM x gemm_reps on the current stream, row pass x row_reps on the side stream, started together."""
    torch.cuda.synchronize()
    e0, e1, s1 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record()
    side.wait_event(e0)
    for _ in range(gemm_reps):
        gemm()
    e1.record()
    with torch.cuda.stream(side):
        for _ in range(row_reps):
            row()
        s1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3, e0.elapsed_time(s1) * 1e3


def main():
    X = ops.split_rows(torch.randn(M, K, device=dev))
    W = ops.split_rows(torch.randn(N, K, device=dev))
    bias = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev)
    z = torch.randn(M, D, device=dev)
    dh = torch.randn(M, D, device=dev)
    gamma, beta = torch.ones(D, device=dev), torch.zeros(D, device=dev)
    mu, rs, _ = ops.ln_prep(z, gamma, beta, 1)
    dzs = torch.empty_like(z)
    gemm = lambda: ops.gemm_split(X, W, bias=bias, out=out)
    rows = {"ln_prep": lambda: ops.ln_prep(z, gamma, beta, 1),
            "ln_act_bwd": lambda: ops.ln_act_bwd(dh, z, mu, rs, gamma, beta, 1, dz_split=dzs, want_dz=False)}
    for f in (gemm, *rows.values()):
        f()
    R = 4
    lib.wf3d_set_option(b"gemm_cus", 0)
    tg = sorted(run(R, 0, gemm, None)[0] for _ in range(3))[1] / R
    print(f"GEMM alone, 256 CUs: {tg:8.1f} us")
    for name, row in rows.items():
        tr = sorted(run(0, R, None, row)[1] for _ in range(3))[1] / R
        print(f"{name} alone: {tr:8.1f} us   -> back to back {tg + tr:8.1f} us per (GEMM + row pass)")
        for g in (256, 240, 224, 208, 192, 160):
            lib.wf3d_set_option(b"gemm_cus", 0 if g == 256 else g)
            ta = sorted(run(R, 0, gemm, None)[0] for _ in range(3))[1] / R
            both = sorted((run(R, R, gemm, row) for _ in range(3)), key=max)[1]
            print(f"  gemm_cus={g:3d}: GEMM alone {ta:8.1f}   together: GEMM stream {both[0] / R:8.1f}  row stream {both[1] / R:8.1f}"
                  f"  -> {max(both) / R:8.1f} us per pair ({(tg + tr) / (max(both) / R):.3f}x)")
        lib.wf3d_set_option(b"gemm_cus", 0)


if __name__ == "__main__":
    main()
