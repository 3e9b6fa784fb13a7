"""Run-time configuration of the HIP path.

precision
    "bf16x3" (default): the large per-point Linear layers (rows >= SPLIT_MIN_ROWS) run as
             split-precision GEMMs — fp32 operands split into bf16 high + low parts, three
             bf16 MFMAs per product, fp32 accumulation: results agree with the fp32 path to
             ~1e-5 (inside the 1e-4 parity gate) at ~3x the fp32-MFMA throughput.
    "fp32":  every GEMM on the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32).
Environment override: WF3D_PRECISION.
"""
import os

SPLIT_MIN_ROWS = 1024
_precision = os.environ.get("WF3D_PRECISION", "bf16x3")
if _precision not in ("fp32", "bf16x3"):
    raise RuntimeError(f"WF3D_PRECISION={_precision!r}: expected 'fp32' or 'bf16x3'")


def precision():
    return _precision


def set_precision(p):
    global _precision
    if p not in ("fp32", "bf16x3"):
        raise ValueError("precision must be 'fp32' or 'bf16x3'")
    _precision = p


# Backward of the edge head: weight / bias gradients that nothing downstream reads ("leaves") are issued on a second
# HIP stream next to the dgrad chain (these launches are 15-60 us kernels that fill a fraction of the chip; in one
# stream each also pays the drain of its predecessor).  WF3D_SIDE_STREAM=0 keeps everything on the current stream.
SIDE_STREAM = os.environ.get("WF3D_SIDE_STREAM", "1") != "0"

# Edge head, bf16x3 mode: the per-vertex Linears (embed, in_proj, out_proj, Pa / Pb: sum-of-counts rows, 14 GFLOP per
# cfg2 step) stay on the exact-fp32 MFMA.  Round 2 ran them on bf16x3 through the general GEMM's x3 staging (0.4 % of
# the step faster); the production-selection frozen-gradient test (tests/test_frozen_grad_gpu.py) then measured 3.2e-4
# on attention.in_proj_weight against 1.7e-4 with fp32 here — the 2e-4 bound on every gradient element is worth more.
# WF3D_EDGE_X3=1 restores the x3 path (kept and tested: tests/test_kernels_gpu.py::test_gemm_x3_layouts).
EDGE_X3 = os.environ.get("WF3D_EDGE_X3", "0") != "0"

# Edge head: store the first edge layer's pre-activation even when backward could rebuild it (A/B switch, WF3D_KEEP_PRE=1).
KEEP_PRE = os.environ.get("WF3D_KEEP_PRE", "0") != "0"

# The stages release their saved activations (2.7 GB at cfg2) at the end of their backward, so a second backward through
# the same graph (retain_graph=True, which the reference's autograd serves) raises.  RETAIN_SAVED = True keeps them
# until the graph itself is freed.
RETAIN_SAVED = os.environ.get("WF3D_RETAIN_SAVED", "0") != "0"
