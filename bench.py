#!/usr/bin/env python3
"""Benchmark of the hot path: point-clouds/sec, forward+backward, synthetic
N=4096x8 clouds (BASELINE.json metric), one process per GPU.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = PointCloudToWireframe forward (train mode, the reference's dropout
p=0.1 active, counts = V for every sample) + backward from a fixed random
cotangent on (vertices, existence_probabilities, edge_probs) [+ the RCCL
gradient all-reduce when N > 1].  Inputs and parameters are resident in HBM
before the timed region; no optimizer step, no Hungarian loss.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "wireframe-3d-prediction_amd"))

import torch  # noqa: E402

CONFIGS = {
    # name: (per-GPU batch, N, V)
    "cfg2": (32, 4096, 64),
    "cfg4": (8, 16384, 64),
    "cfg5": (32, 4096, 256),
    "cfg1": (1, 1024, 32),
}
FP32_MFMA_PEAK_TFLOPS = 157.3       # MI355X_MICROARCH.md, chip-level parameters
BF16_MFMA_PEAK_TFLOPS = 2500.0      # dense bf16 MFMA (never the 2:1-sparsity figure)
HBM_PEAK_GBS = 8000.0               # HBM3E spec (6.3 TB/s measured on a float4 copy, same guide)
PROFILE_TAG = "r03"                 # rocprofv3 summaries of this round's build under profiles/
SEED = 1234


def algorithmic_flops_per_cloud(N, V, hidden=(512, 1024, 2048, 1024), out=512, din=8):
    """Reference-formulation FLOPs of fwd+bwd per cloud:
    MAC x 2, backward = 2 x forward."""
    dims = [din] + list(hidden) + [out]
    enc = N * sum(a * b for a, b in zip(dims[:-1], dims[1:]))
    fusion = 1024 * 2048 + 2048 * 1024 + 1024 * 512
    vert = 1024 * 512 + 512 * 4096 + 4096 * 2048 + 2048 * 2048 + 2048 * 1024 + 1024 * 4 * V + 512 * 2048 + 512 * 1024
    E = V * (V - 1) // 2
    edge = V * (3 * 256 + 256 * 512 + 512 * 1536 + 512 * 512) + 2 * V * V * 512 \
        + E * (1031 * 512 + 512 * 256 + 256 * 128 + 128)
    return 6 * (enc + fusion + vert + edge), 6 * enc


def make_inputs(B, N, V, rank, device):
    g = torch.Generator().manual_seed(SEED + 7919 * rank)     # per global sample shard
    x = torch.randn(B, N, 8, generator=g)
    counts = torch.full((B,), V, dtype=torch.long)
    E = V * (V - 1) // 2
    cot = {"vertices": torch.randn(B, V, 3, generator=g),
           "existence_probabilities": torch.randn(B, V, generator=g),
           "edge_probs": torch.randn(B, E, generator=g)}
    return x.to(device), counts.to(device), {k: v.to(device) for k, v in cot.items()}


class OpTimer:
    """HIP-event timing of individual C-ABI calls inside the timed region, on the stream they are launched on (torch's
    current stream).  Every wrapped call is bracketed by two events; `work(args, result)` gives its algorithmic
    (flops, bytes).  Groups:
      gemm   ops.gemm_split     forward + dgrad GEMMs of the per-point MLP  -> gemm_split_x16p_kernel (the dominant kernel)
      wgrad  ops.gemm_split_tn  weight-gradient GEMMs                        -> gemm_split_x16_kernel<true> + split_reduce
      rows   ops.ln_prep / ln_act_bwd / first_layer_fwd / ln_act_bwd_first   LayerNorm passes between the GEMMs (HBM-bound)
      pool   ops.pool4_fwd / pool4_bwd                                       the 4-way pool and its scatter (HBM-bound)
    Only launches above `min_flops` / `min_bytes` are timed (the encoder's; the heads' small ones are left alone)."""

    def __init__(self, min_flops=1e11, min_bytes=1e8):
        self.min_flops, self.min_bytes, self.recs, self._saved = min_flops, min_bytes, [], []
        self.active, self.steps_sampled = False, 0        # the ~70 events of a sampled step cost ~1 % of it: every 4th step is sampled

    def _wrap(self, mod, name, group, work):
        fn = getattr(mod, name)
        timer = self

        def timed(*a, **kw):
            if not timer.active:
                return fn(*a, **kw)
            fl, by = work(a, kw)
            if fl < timer.min_flops and by < timer.min_bytes:
                return fn(*a, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*a, **kw)
            e1.record()
            timer.recs.append((group, name, e0, e1, fl, by))
            return out

        self._saved.append((mod, name, fn))
        setattr(mod, name, timed)

    def install(self, ops, split):
        el = lambda t: t.shape[0] * t.shape[1]                                     # noqa: E731
        if split:
            self._wrap(ops, "gemm_split", "gemm", lambda a, k: (2.0 * a[0].shape[0] * a[1].shape[0] * a[0].shape[1],
                                                                4.0 * (el(a[0]) + el(a[1]) + a[0].shape[0] * a[1].shape[0])))
            self._wrap(ops, "gemm_split_tn", "wgrad", lambda a, k: (2.0 * a[0].shape[0] * a[0].shape[1] * a[1].shape[1],
                                                                    4.0 * (el(a[0]) + el(a[1]) + a[0].shape[1] * a[1].shape[1])))
        else:
            def gw(a, k):
                x, w, layout = a[0], a[1], a[2]
                if layout == ops.TN:
                    M, N, K = x.shape[1], w.shape[1], x.shape[0]
                elif layout == ops.NN:
                    M, N, K = x.shape[0], w.shape[1], x.shape[1]
                else:
                    M, N, K = x.shape[0], w.shape[0], x.shape[1]
                return 2.0 * M * N * K, 4.0 * (M * K + N * K + M * N)
            self._wrap(ops, "gemm", "gemm", gw)
        # row passes: algorithmic bytes per element: z read + h written; dh, z read + dz written
        self._wrap(ops, "ln_prep", "rows", lambda a, k: (0.0, 8.0 * el(a[0])))
        self._wrap(ops, "ln_act_bwd", "rows", lambda a, k: (0.0, (12.0 if k.get("dz_split") is None or not k.get("want_dz", True) else 16.0) * el(a[1])))
        self._wrap(ops, "first_layer_fwd", "rows", lambda a, k: (0.0, 8.0 * a[0].shape[0] * a[1].shape[0]))
        self._wrap(ops, "ln_act_bwd_first", "rows", lambda a, k: (0.0, 8.0 * el(a[1])))
        self._wrap(ops, "pool4_fwd", "pool", lambda a, k: (0.0, 4.0 * a[0].numel()))
        self._wrap(ops, "pool4_bwd", "pool", lambda a, k: (0.0, 4.0 * a[9] * a[10] * a[11]))

    def uninstall(self):
        for mod, name, fn in reversed(self._saved):
            setattr(mod, name, fn)
        self._saved = []

    def summary(self):
        """group -> dict(launches, ms, flops, bytes, per-op ms)"""
        out = {}
        for group, name, e0, e1, fl, by in self.recs:
            g = out.setdefault(group, {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0, "ops": {}})
            ms = e0.elapsed_time(e1)
            g["launches"] += 1; g["ms"] += ms; g["flops"] += fl; g["bytes"] += by
            o = g["ops"].setdefault(name, [0, 0.0, 0.0])
            o[0] += 1; o[1] += ms; o[2] += by
        return out


def host_cores():
    """Cores this process may actually use: affinity mask, capped by the cgroup
    CPU quota (a 1-GPU box gives a 16-core share of a 256-thread host)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
            break
        except (OSError, ValueError, IndexError):
            continue
    cap = os.environ.get("WF3D_CPU_THREADS")          # optional explicit cap (unset: every core the process may use)
    return min(n, int(cap)) if cap else n


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(model, cfg, N, V, seconds_budget=25.0):
    """The CPU oracle (oracle/reference_cpu.py, proven equal to the reference in
    the build container) timed on this box's host cores on a bounded sample of
    the same workload: the benchmarked config's shape (N points, V vertices) at a reduced batch
    (~1 s of CPU work per step), 1 warm-up + up to 3 timed steps."""
    from oracle import reference_cpu as oracle
    cores = host_cores()
    torch.set_num_threads(cores)
    Bc = {"cfg4": 1, "cfg5": 2, "cfg1": 1}.get(cfg, 4)
    P = oracle.params_from_module(model)
    g = torch.Generator().manual_seed(SEED)
    x = torch.randn(Bc, N, 8, generator=g)
    counts = torch.full((Bc,), V, dtype=torch.long)
    E = V * (V - 1) // 2
    cot = {"vertices": torch.randn(Bc, V, 3, generator=g), "existence_probabilities": torch.randn(Bc, V, generator=g),
           "edge_probs": torch.randn(Bc, E, generator=g)}
    times = []
    t_start = time.time()
    for it in range(4):
        for p in P.values():
            p.grad = None
        t0 = time.time()
        out = oracle.model_forward(P, x, counts, V, training=True)
        sum((out[k] * cot[k]).sum() for k in cot).backward()
        dt = time.time() - t0
        if it > 0:
            times.append(dt)
        if time.time() - t_start > seconds_budget and times:
            break
    med = statistics.median(times)
    return {"value": Bc / med, "unit": "clouds/s", "cores": cores, "cpu_model": cpu_model(), "kind": "port",
            "sample": f"oracle/reference_cpu.py fwd+bwd, {cfg} shape N={N} V={V} at batch {Bc}, torch intra-op threads = {cores} "
                      f"(affinity mask capped by the cgroup CPU quota), 1 warm-up + {len(times)} timed steps, "
                      f"median {med * 1e3:.0f} ms/step, dropout 0"}


def pmc_traffic(cfg):
    """{"traffic": bytes beyond L2 per dominant-kernel launch, "traffic_profile": where it was measured} from the committed
    rocprofv3 counter summary of this config (profiles/, scripts/summarize_profiles.py); None if there is none."""
    name = f"{PROFILE_TAG}_pmc_summary.json" if cfg == "cfg2" else f"{PROFILE_TAG}_{cfg}_pmc_summary.json"
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", name)
    try:
        with open(path) as f:
            d = json.load(f).get("cfg2" if cfg == "cfg2" else cfg) or {}
        return {"traffic": d.get("gemm_hbm_bytes_per_launch"),
                "traffic_profile": f"profiles/{name}: FETCH_SIZE x 2 + WRITE_SIZE per launch ({d.get('gemm_read_bytes_per_launch', 0) / 1e6:.0f} MB read + "
                                   f"{d.get('gemm_write_bytes_per_launch', 0) / 1e6:.0f} MB written), separate --pmc passes of this bench"}
    except (OSError, ValueError):
        return {"traffic": None, "traffic_profile": f"profiles/{name} not found"}


def power_probe(ops, device, seconds=1.2):
    """The dominant GEMM shape (131072 x 1024 -> 2048) in a loop under the board-power sensor, and the vendor library's
    plain bf16 GEMM of the same shape beside it: what the MFMA pipes deliver under this board's power cap."""
    from wf3d import telemetry
    hw = telemetry.hwmon_dir(device.index or 0)
    M, K, N = 131072, 1024, 2048
    x, w = torch.randn(M, K, device=device), torch.randn(N, K, device=device)
    X, W = ops.split_rows(x), ops.split_rows(w)
    out = torch.empty(M, N, device=device)
    fl = 2.0 * M * K * N
    t, watts, ghz = telemetry.run_sampled(lambda: ops.gemm_split(X, W, out=out), hw, seconds)
    xb, wb = x.bfloat16(), w.bfloat16()
    ob = torch.empty(M, N, device=device, dtype=torch.bfloat16)
    del x, w
    tl, lwatts, lghz = telemetry.run_sampled(lambda: torch.matmul(xb, wb.t(), out=ob), hw, seconds)
    return {"shape": f"{M} x {K} -> {N}, normal random operands", "board_w": watts, "cap_w": telemetry.power_cap_watts(hw),
            "sclk_ghz": ghz, "us": t * 1e6, "executed_mfma_tflops": 3.0 * fl / t / 1e12,
            "library_bf16_gemm": {"what": "torch.matmul(bf16, bf16) of the same shape (hipBLASLt), one MFMA per product",
                                  "us": tl * 1e6, "mfma_tflops": fl / tl / 1e12, "board_w": lwatts, "sclk_ghz": lghz},
            "sensor": hw or "hwmon not visible (power fields null)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-power-probe", action="store_true", help="skip the board-power / library-GEMM yardstick after the timed region")
    ap.add_argument("--no-op-timers", action="store_true", help="diagnostic: no per-op HIP events (roofline blocks become empty)")
    ap.add_argument("--dropout", type=float, default=0.1)
    ap.add_argument("--precision", choices=["fp32", "bf16x3"], default=None,
                    help="arithmetic of the large GEMMs (default: wf3d.config / WF3D_PRECISION = bf16x3)")
    args = ap.parse_args()

    from wf3d import config
    from wf3d import dist as wd
    from wf3d import ops
    from models.PointCloudToWireframe import PointCloudToWireframe
    if args.precision:
        config.set_precision(args.precision)
    split = config.precision() == "bf16x3"

    rank, world, device = wd.init_from_env("cuda")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run")
    B, N, V = CONFIGS[args.config]

    torch.manual_seed(SEED)
    model = PointCloudToWireframe(input_dim=8, max_vertices=V).to(device)
    model.vertex_predictor.ensure_point_pool_proj(1024, device)       # lazy layer, identical on every rank
    wd.sync_parameters(model)
    model.set_dropout(args.dropout)
    model.train()
    reducer = wd.GradReducer(model) if world > 1 else None
    x, counts, cot = make_inputs(B, N, V, rank, device)
    inv = 1.0 / (B * world)
    keys = list(cot)
    cots = [cot[k] * inv for k in keys]      # d/dtheta of loss = sum_k <out_k, c_k> / B_global, fed to autograd directly

    def step():
        model.zero_grad(set_to_none=True)
        out = model(x, counts)
        torch.autograd.backward([out[k] for k in keys], cots)
        if reducer is not None:
            reducer.finish()

    def fence():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # all process-group setup happened in init_from_env(), before any kernel of ours ran on this rank
    for _ in range(args.warmup):
        step()
    if reducer is not None:
        reducer.exposed_ms()                  # drop the warm-up records
    timer = OpTimer()
    if not args.no_op_timers:
        timer.install(ops, split)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    fence()
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        timer.active = (i % 4 == 0) and not args.no_op_timers
        timer.steps_sampled += int(timer.active)
        step()
        marks[i + 1].record()
    timer.active = False
    fence()
    dt = time.perf_counter() - t0
    timer.uninstall()
    t = torch.tensor([dt], dtype=torch.float64, device=device)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    dt = float(t.item())

    if rank == 0:
        groups = timer.summary()
        ns = max(timer.steps_sampled, 1)                  # steps in which the per-op events were recorded
        step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
        total_fl, enc_fl = algorithmic_flops_per_cloud(N, V)
        g = groups.get("gemm", {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
        achieved = g["flops"] / (g["ms"] * 1e-3) / 1e12 if g["ms"] > 0 else 0.0
        if split:
            # bf16x3: every algorithmic product costs three bf16 MFMA products (hi*hi + hi*lo + lo*hi)
            peak, kern = BF16_MFMA_PEAK_TFLOPS, "gemm_split_x16p_kernel (persistent, 256x256x32 tile, LDS-DMA staged, 3 x v_mfma_f32_16x16x32_bf16 per product)"
            extra = {"executed_mfma_tflops": 3.0 * achieved, "executed_frac": 3.0 * achieved / peak,
                     "note": "achieved = algorithmic 2MNK FLOP / time; the split algorithm issues 3 MFMA FLOP per algorithmic FLOP"}
        else:
            peak, kern = FP32_MFMA_PEAK_TFLOPS, "gemm_kernel<2,2,2,2> (128x128x32 v_mfma_f32_32x32x2_f32)"
            extra = {}
        roof = {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                # HBM-side bytes come from rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, own runs), not from this run:
                # the figure of the committed summary for this config, per launch like `achieved`
                **pmc_traffic(args.config),
                "kernel": kern, "launches_timed": g["launches"], "algorithmic_bytes_per_launch": g["bytes"] / max(g["launches"], 1),
                "gemm_ms_per_step": g["ms"] / ns, "steps_sampled": timer.steps_sampled,
                "whole_step_tflops": total_fl * B * world * args.steps / dt / 1e12, **extra}
        if "wgrad" in groups:
            w = groups["wgrad"]
            wa = w["flops"] / (w["ms"] * 1e-3) / 1e12
            roof["wgrad"] = {"kernel": "gemm_split_x16_kernel<true> (+ split_reduce_kernel): dW = dz^T h on reduction-major sx8 operands",
                             "bound": "mfma", "achieved": wa, "peak": peak, "unit": "TFLOP/s", "frac": wa / peak,
                             "launches_timed": w["launches"], "ms_per_step": w["ms"] / ns}
        hbm = {}
        for grp, label in (("rows", "LayerNorm passes between the GEMMs"), ("pool", "4-way pool + its scatter")):
            if grp in groups:
                r = groups[grp]
                gbs = r["bytes"] / (r["ms"] * 1e-3) / 1e9
                hbm[grp] = {"what": label, "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": gbs / HBM_PEAK_GBS, "ms_per_step": r["ms"] / ns,
                            "algorithmic_bytes_per_step": r["bytes"] / ns,
                            "ops": {k: {"launches_per_step": v[0] / ns, "ms_per_step": v[1] / ns,
                                        "GB/s": v[2] / (v[1] * 1e-3) / 1e9} for k, v in r["ops"].items()}}
        roof["hbm"] = hbm
        res = {
            "metric": "point-clouds/sec fwd+bwd",
            "value": B * world * args.steps / dt,
            "unit": "clouds/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "ms_per_step_median_hip_events": step_ms[len(step_ms) // 2],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16x3" if split else "f32", "data": "synthetic",
            "config": {"workload": f"{args.config}: N={N}x8 clouds, max_vertices={V}, batch {B}/GPU, train-mode fwd+bwd, "
                                   f"edge-head dropout p={args.dropout}, counts=V",
                       "global_batch": B * world, "num_points": N, "max_vertices": V,
                       "parallelism": f"dp{world}", "algorithmic_gflop_per_cloud": total_fl / 1e9,
                       "arithmetic": ("fp32 operands split into bf16 hi+lo, 3 bf16 MFMAs per product, fp32 accumulate "
                                      "(forward outputs within 1e-4 of the fp32 reference element-wise; every parameter-gradient element within 2e-4 of "
                                      "max(|g|, rms g) of an fp64 oracle under the kernels' own ReLU / arg-max decisions, measured worst 1.7e-4 at the "
                                      "production kernel selection, tests/test_frozen_grad_gpu.py; fp32 mode: 1e-4, measured 1.8e-5): per-point MLP and "
                                      "the two wide edge-MLP layers; per-vertex edge-head Linears, M=batch-row head Linears, K<=8 products and attention on fp32 MFMA")
                       if split else "fp32 MFMA everywhere"},
            "roofline": roof,
        }
        if world > 1:
            ex = sorted(reducer.exposed_ms())
            res["rccl_ranks"] = torch.distributed.get_world_size()
            res["dist_backend"] = torch.distributed.get_backend()
            res["allreduce_exposed_ms"] = ex[len(ex) // 2] if ex else None
            res["allreduce_buckets_mb"] = [round(nb / 2 ** 20, 1) for _, nb in reducer.bucket_summary()]
        # the two side measurements must never cost the bench line itself
        if world == 1 and split and not args.no_power_probe:
            try:
                roof["power"] = power_probe(ops, device)
            except Exception as e:  # noqa: BLE001
                roof["power"] = {"error": repr(e)[:300]}
        if world == 1 and not args.no_cpu_baseline:
            try:
                res["cpu_baseline"] = cpu_baseline(model, args.config, N, V)
            except Exception as e:  # noqa: BLE001
                res["cpu_baseline"] = {"value": None, "unit": "clouds/s", "cores": host_cores(), "kind": "port", "sample": "failed: " + repr(e)[:300]}
        print(json.dumps(res), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
