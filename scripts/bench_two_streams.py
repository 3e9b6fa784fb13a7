#!/usr/bin/env python3
"""Would two half-batches on two streams fill the bubbles of one full batch?  cfg2's 32 clouds as one model call on one
stream, against 2 x 16 clouds on two streams (two model copies: timing only), and 2 x 16 back to back on one stream."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wireframe-3d-prediction_amd"))
sys.path.insert(0, ROOT)
from models.PointCloudToWireframe import PointCloudToWireframe  # noqa: E402
from wf3d import _lib  # noqa: E402

dev = torch.device("cuda:0")
N, V = 4096, 64


def make(B, seed):
    torch.manual_seed(seed)
    m = PointCloudToWireframe(input_dim=8, max_vertices=V).to(dev)
    m.vertex_predictor.ensure_point_pool_proj(1024, dev)
    m.set_dropout(0.1)
    m.train()
    g = torch.Generator(device=dev).manual_seed(seed)
    x = torch.randn(B, N, 8, generator=g, device=dev)
    counts = torch.full((B,), V, dtype=torch.long, device=dev)
    out = m(x, counts)
    keys = ["vertices", "existence_probabilities", "edge_probs"]
    cots = [torch.randn(out[k].shape, generator=g, device=dev) / B for k in keys]
    return m, x, counts, keys, cots


def step(pack):
    m, x, counts, keys, cots = pack
    m.zero_grad(set_to_none=True)
    out = m(x, counts)
    torch.autograd.backward([out[k] for k in keys], cots)


def timed(fn, reps=12, warm=4):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    full = make(32, 1)
    ha, hb = make(16, 2), make(16, 3)
    sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    lib = _lib.load()

    def two_streams():
        with torch.cuda.stream(sa):
            step(ha)
        with torch.cuda.stream(sb):
            step(hb)

    def two_streams_interleaved():
        # forward A, forward B, backward A, backward B: the host enqueues the halves' phases alternately
        for pack, s in ((ha, sa), (hb, sb)):
            with torch.cuda.stream(s):
                pack[0].zero_grad(set_to_none=True)
        outs = []
        for pack, s in ((ha, sa), (hb, sb)):
            with torch.cuda.stream(s):
                outs.append(pack[0](pack[1], pack[2]))
        for (pack, s), out in zip(((ha, sa), (hb, sb)), outs):
            with torch.cuda.stream(s):
                torch.autograd.backward([out[k] for k in pack[3]], pack[4])

    def back_to_back():
        step(ha)
        step(hb)

    t_full = timed(lambda: step(full))
    print(f"one call, 32 clouds, one stream:           {t_full:7.2f} ms  ({32 / t_full * 1e3:7.0f} clouds/s)")
    t_bb = timed(back_to_back)
    print(f"2 x 16 clouds back to back, one stream:    {t_bb:7.2f} ms  ({32 / t_bb * 1e3:7.0f} clouds/s)")
    for cus in (0, 224, 192):
        lib.wf3d_set_option(b"gemm_cus", cus)
        t2 = timed(two_streams)
        t3 = timed(two_streams_interleaved)
        print(f"2 x 16 clouds on two streams, gemm_cus={cus or 256}: {t2:7.2f} ms  ({32 / t2 * 1e3:7.0f} clouds/s);  phases interleaved: {t3:7.2f} ms  ({32 / t3 * 1e3:7.0f} clouds/s)")
    lib.wf3d_set_option(b"gemm_cus", 0)


if __name__ == "__main__":
    main()
