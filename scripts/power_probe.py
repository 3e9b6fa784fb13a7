#!/usr/bin/env python3
"""Board power and shader clock (wf3d.telemetry: the card's hwmon files) while one kernel runs in a loop, with the
vendor library's bf16 GEMM on the same shapes as a yardstick for what the 1,400 W cap lets the MFMA pipes do, and the
same kernels on an all-zero activation operand (no toggling in the multipliers) to show the cap is what binds:
  python scripts/power_probe.py      -> profiles/r03_power.txt"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wireframe-3d-prediction_amd"))
from wf3d import ops, telemetry  # noqa: E402

dev = torch.device("cuda:0")
M = 131072


def loop(name, fn, hw, flops):
    el, watts, ghz = telemetry.run_sampled(fn, hw, seconds=3.0)
    rate = f"  {flops / el / 1e12:7.1f} TFLOP/s of MFMA work" if flops else ""
    print(f"{name:48s} {el * 1e6:8.1f} us  {watts or float('nan'):7.0f} W  {ghz or float('nan'):5.2f} GHz{rate}", flush=True)


def main():
    hw = telemetry.hwmon_dir(0)
    print(f"sensor {hw}  power cap {telemetry.power_cap_watts(hw)} W")
    for K, N in ((1024, 2048), (2048, 1024), (512, 1024)):
        x, w, g = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev), torch.randn(M, N, device=dev)
        X, W, G = ops.split_rows(x), ops.split_rows(w), ops.split_rows(g)
        out, dw = torch.empty(M, N, device=dev), torch.empty(N, K, device=dev)
        f = 2.0 * M * K * N
        loop(f"bf16x3 forward {K}->{N} (3 MFMA per product)", lambda: ops.gemm_split(X, W, out=out), hw, 3 * f)
        loop(f"bf16x3 wgrad {N}x{K}", lambda: ops.gemm_split_tn(G, X, out=dw), hw, 3 * f)
        Xz = torch.zeros_like(X)
        loop(f"bf16x3 forward {K}->{N}, activations = 0", lambda: ops.gemm_split(Xz, W, out=out), hw, 3 * f)
        xb, wb, gb = x.bfloat16(), w.bfloat16(), g.bfloat16()
        ob = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        loop(f"library bf16 forward {K}->{N} (x @ w.T)", lambda: torch.matmul(xb, wb.t(), out=ob), hw, f)
        loop(f"library bf16 wgrad {N}x{K} (g.T @ x)", lambda: torch.matmul(gb.t(), xb), hw, f)
        xz = torch.zeros_like(xb)
        loop(f"library bf16 forward {K}->{N}, activations = 0", lambda: torch.matmul(xz, wb.t(), out=ob), hw, f)
        del x, w, g, X, W, G, out, dw, Xz, xb, wb, gb, ob, xz
    z = torch.randn(M, 1024, device=dev)
    dh = torch.randn(M, 1024, device=dev)
    gm, b = torch.ones(1024, device=dev), torch.zeros(1024, device=dev)
    mu, rs, _ = ops.ln_prep(z, gm, b, 1)
    dzs = torch.empty_like(z)
    loop("ln_prep 131072 x 1024", lambda: ops.ln_prep(z, gm, b, 1), hw, 0)
    loop("ln_act_bwd 131072 x 1024", lambda: ops.ln_act_bwd(dh, z, mu, rs, gm, b, 1, dz_split=dzs, want_dz=False), hw, 0)


if __name__ == "__main__":
    main()
