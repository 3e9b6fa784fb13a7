#!/usr/bin/env python3
"""Turn the raw rocprofv3 CSVs that a gpurun call left under gpurun_out/ into the
committed summaries under profiles/ (round-tagged), and into
profiles/pmc_summary.json, which bench.py reads for `roofline.traffic`.

    python scripts/summarize_profiles.py r02            # cfg2 (bench.py default)
    python scripts/summarize_profiles.py r02_cfg5 cfg5  # another config: dirs gpurun_out/prof_r02_cfg5_*

Inputs (written on the GPU box by the commands recorded in profiles/README.md):
    gpurun_out/prof_<tag>_trace*/**/_kernel_stats.csv   rocprofv3 --kernel-trace --stats
    gpurun_out/prof_<tag>_fetch/**/_counter_collection.csv   --pmc FETCH_SIZE      (own pass)
    gpurun_out/prof_<tag>_write/**/_counter_collection.csv   --pmc WRITE_SIZE      (own pass)
    gpurun_out/prof_<tag>_sq/**/_counter_collection.csv      --pmc SQ_* GRBM_GUI_ACTIVE

HBM-byte corrections (MI355X_MICROARCH.md §HBM): FETCH_SIZE/WRITE_SIZE are in KiB;
on gfx950 FETCH_SIZE counts 64 B per 128-B request for 16-B-per-lane coalesced
loads, which is how the GEMM stages both operands -> read bytes = 2 x FETCH_SIZE x 1024.
WRITE_SIZE is exact for the 4-B-per-lane / 16-B-per-lane stores used here.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
cfg = sys.argv[2] if len(sys.argv) > 2 else "cfg2"
CFG_DESC = {"cfg2": "B=32, N=4096, V=64", "cfg4": "B=8, N=16384, V=64", "cfg5": "B=32, N=4096, V=256"}[cfg]
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
os.makedirs(P, exist_ok=True)


def one(pattern):
    fs = glob.glob(os.path.join(G, pattern), recursive=True)
    return max(fs, key=os.path.getmtime) if fs else None      # the latest run (process ids in the names do not sort by time)


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:70]


# ---- kernel-trace stats -------------------------------------------------------
ks = one(f"prof_{tag}_trace*/**/*_kernel_stats.csv")
stats_rows = []
if ks:
    shutil.copy(ks, os.path.join(P, f"{tag}_kernel_stats.csv"))
    rows = list(csv.DictReader(open(ks)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    for r in rows:
        stats_rows.append((short(r["Name"]), int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6,
                           float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))


def pmc(pattern):
    f = one(pattern)
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    if not f:
        return out
    for r in csv.DictReader(open(f)):
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        out[short(r["Kernel_Name"])][r["Counter_Name"]].append((float(r["Counter_Value"]), dur, int(r["Dispatch_Id"])))
    return out


fetch, write, sq = pmc(f"prof_{tag}_fetch/**/*_counter_collection.csv"), \
    pmc(f"prof_{tag}_write/**/*_counter_collection.csv"), pmc(f"prof_{tag}_sq/**/*_counter_collection.csv")

BIG_US = 300.0          # the encoder's twelve >=137-GFLOP GEMM launches per step all run > 0.35 ms
DOM = os.environ.get("WF3D_DOMINANT", "gemm_split_x16p_kernel")   # dominant kernel (fp32 mode: "gemm_kernel<2, 2, 2, 2")


def big(vals):
    return [v for v, d, _ in vals if d >= BIG_US]


summary = {"tag": tag, "kernels": {}}
gemm_fetch, gemm_write = [], []
for name in sorted(set(fetch) | set(write)):
    fv = fetch.get(name, {}).get("FETCH_SIZE", [])
    wv = write.get(name, {}).get("WRITE_SIZE", [])
    e = {"launches": len(fv) or len(wv)}
    if fv:
        e["read_bytes_avg"] = 2 * 1024 * sum(v for v, _, _ in fv) / len(fv)
    if wv:
        e["write_bytes_avg"] = 1024 * sum(v for v, _, _ in wv) / len(wv)
    summary["kernels"][name] = e
    if name.startswith(DOM):
        gemm_fetch += big(fv)
        gemm_write += big(wv)
mfma = {}
for name, ctrs in sq.items():
    if not (name.startswith(DOM) or name.startswith("gemm_split_x16_kernel")):
        continue
    busy = sum(v for v, d, _ in ctrs.get("SQ_VALU_MFMA_BUSY_CYCLES", []) if d >= BIG_US)
    gui = sum(v for v, d, _ in ctrs.get("GRBM_GUI_ACTIVE", []) if d >= BIG_US)
    waves = sum(v for v, d, _ in ctrs.get("SQ_WAVE_CYCLES", []) if d >= BIG_US)
    sqb = sum(v for v, d, _ in ctrs.get("SQ_BUSY_CYCLES", []) if d >= BIG_US)
    mops = sum(v for v, d, _ in ctrs.get("SQ_INSTS_VALU_MFMA_MOPS_F32", []) if d >= BIG_US)
    n = len([1 for v, d, _ in ctrs.get("GRBM_GUI_ACTIVE", []) if d >= BIG_US])
    dur_us = sum(d for v, d, _ in ctrs.get("GRBM_GUI_ACTIVE", []) if d >= BIG_US)
    if n:
        mfma[name] = {"launches": n, "SQ_VALU_MFMA_BUSY_CYCLES": busy, "GRBM_GUI_ACTIVE": gui, "SQ_WAVE_CYCLES": waves,
                      "SQ_BUSY_CYCLES": sqb, "SQ_INSTS_VALU_MFMA_MOPS_F32": mops,
                      # GRBM_GUI_ACTIVE is summed over the 8 XCDs; MFMA_BUSY over all 1024 SIMDs' pipes
                      "mfma_busy_frac": busy / (gui / 8.0 * 1024.0) if gui else None,
                      # effective clock under load (MI355X_MICROARCH.md, DVFS give-back): GRBM_GUI_ACTIVE / 8 / wall
                      "effective_clock_ghz": gui / 8.0 / (dur_us * 1e3) if dur_us else None}
summary["gemm_mfma"] = mfma
# every kernel: MFMA pipe occupancy over all its launches, and HBM-side GB/s from the two byte passes
per_kernel = {}
for name, ctrs in sq.items():
    busy = sum(v for v, d, _ in ctrs.get("SQ_VALU_MFMA_BUSY_CYCLES", []))
    gui = sum(v for v, d, _ in ctrs.get("GRBM_GUI_ACTIVE", []))
    dur = sum(d for v, d, _ in ctrs.get("GRBM_GUI_ACTIVE", []))
    n = len(ctrs.get("GRBM_GUI_ACTIVE", []))
    if n and gui:
        per_kernel[name] = {"launches": n, "avg_us": dur / n, "mfma_busy_frac": busy / (gui / 8.0 * 1024.0),
                            "effective_clock_ghz": gui / 8.0 / (dur * 1e3)}
for name in set(fetch) | set(write):
    fv = fetch.get(name, {}).get("FETCH_SIZE", [])
    wv = write.get(name, {}).get("WRITE_SIZE", [])
    e = per_kernel.setdefault(name, {})
    if fv:
        e["read_bytes_avg"] = 2 * 1024 * sum(v for v, _, _ in fv) / len(fv)
        e["read_pass_avg_us"] = sum(d for _, d, _ in fv) / len(fv)
    if wv:
        e["write_bytes_avg"] = 1024 * sum(v for v, _, _ in wv) / len(wv)
        e["write_pass_avg_us"] = sum(d for _, d, _ in wv) / len(wv)
    us = [e[k] for k in ("read_pass_avg_us", "write_pass_avg_us") if k in e]
    if us:
        e["hbm_GBps"] = (e.get("read_bytes_avg", 0.0) + e.get("write_bytes_avg", 0.0)) / (sum(us) / len(us) * 1e-6) / 1e9
summary["per_kernel"] = per_kernel
summary["config"] = cfg
if gemm_fetch and gemm_write:
    rd = 2 * 1024 * sum(gemm_fetch) / len(gemm_fetch)
    wr = 1024 * sum(gemm_write) / len(gemm_write)
    summary[cfg] = {"kernel": DOM, "gemm_launches": len(gemm_fetch), "gemm_read_bytes_per_launch": rd,
                       "gemm_write_bytes_per_launch": wr, "gemm_hbm_bytes_per_launch": rd + wr,
                       "correction": "read = 2 x FETCH_SIZE KiB (gfx950 16-B/lane streams), write = WRITE_SIZE KiB"}
json.dump(summary, open(os.path.join(P, f"{tag}_pmc_summary.json"), "w"), indent=1)

with open(os.path.join(P, f"{tag}_summary.md"), "w") as f:
    f.write(f"# rocprofv3 summary, {tag} ({cfg}: {CFG_DESC}; dominant kernel `{DOM}`)\n\n")
    f.write("## kernel-trace --stats (3 timed + 1 warm-up steps)\n\n| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
    for n, c, t, a, pc in stats_rows[:30]:
        f.write(f"| `{n}` | {c} | {t:.2f} | {a:.1f} | {pc:.1f} |\n")
    if cfg in summary:
        c = summary[cfg]
        f.write(f"\n## HBM-side traffic of the dominant kernel (`{DOM}`, launches >= {BIG_US:.0f} us)\n\n")
        f.write(f"launches {c['gemm_launches']}: read {c['gemm_read_bytes_per_launch'] / 1e6:.1f} MB + write "
                f"{c['gemm_write_bytes_per_launch'] / 1e6:.1f} MB = {c['gemm_hbm_bytes_per_launch'] / 1e6:.1f} MB per launch "
                f"({c['correction']})\n")
    if mfma:
        f.write("\n## MFMA pipe occupancy of the dominant kernel (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x 1024 SIMDs))\n\n")
        for n, e in mfma.items():
            f.write(f"* `{n}`: {e['launches']} launches, busy fraction {e['mfma_busy_frac']:.3f}, effective clock "
                    f"{e['effective_clock_ghz']:.2f} GHz (GRBM_GUI_ACTIVE / 8 / wall)\n")
    # ---- whole step: bytes beyond L2 against the algorithmic minimum ----
    PASS_STEPS = 2                                   # the byte passes run `--steps 1 --warmup 1`
    tot_rd = sum(2 * 1024 * v for k in fetch.values() for v, _, _ in k.get("FETCH_SIZE", [])) / PASS_STEPS
    tot_wr = sum(1024 * v for k in write.values() for v, _, _ in k.get("WRITE_SIZE", [])) / PASS_STEPS
    Bc, Nc, Vc = {"cfg2": (32, 4096, 64), "cfg4": (8, 16384, 64), "cfg5": (32, 4096, 256)}[cfg]
    act = Bc * Nc * (8 + 512 + 1024 + 2048 + 1024) * 4          # pre-activations kept for backward, fp32
    pf = Bc * Nc * 512 * 4                                      # point_features (returned by PointNetEncoder.forward)
    alg = 2 * act + 2 * pf + 3 * 124.2e6                        # written once + read once; parameters fwd + bwd + gradient write
    if tot_rd or tot_wr:
        summary["whole_step"] = {"read_bytes": tot_rd, "write_bytes": tot_wr, "algorithmic_bytes": alg,
                                 "ratio": (tot_rd + tot_wr) / alg}
        json.dump(summary, open(os.path.join(P, f"{tag}_pmc_summary.json"), "w"), indent=1)
        f.write(f"\n## whole step: bytes beyond L2\n\nsum over all kernels of FETCH_SIZE x 2 + WRITE_SIZE = {tot_rd / 1e9:.1f} + {tot_wr / 1e9:.1f} = "
                f"**{(tot_rd + tot_wr) / 1e9:.1f} GB per step**; algorithmic minimum with stored pre-activations (SURVEY.md section 8d: "
                f"2 x {act / 1e9:.2f} GB activations + 2 x {pf / 1e9:.2f} GB point_features + 3 x 0.124 GB parameters) = {alg / 1e9:.1f} GB "
                f"-> ratio {(tot_rd + tot_wr) / alg:.1f}x\n")
    top = sorted(((e.get("avg_us", 0.0) * e.get("launches", 0), n, e) for n, e in per_kernel.items()), reverse=True)
    top = [t for t in top if t[0] > 0.0]             # every kernel that ran (no cut: cfg5 must show the attention kernels)
    if top:
        f.write("\n## per kernel: MFMA pipe occupancy (SQ pass) and HBM-side bytes per launch (FETCH_SIZE x 2 x 1 KiB, WRITE_SIZE x 1 KiB; own passes)\n\n")
        f.write("| kernel | launches | avg us | MFMA busy | clock GHz | read MB | write MB | GB/s |\n|---|---|---|---|---|---|---|---|\n")
        for _, n, e in top:
            fmt = lambda k, sc=1.0, nd=1: (f"{e[k] * sc:.{nd}f}" if k in e and e[k] is not None else "-")   # noqa: E731
            f.write(f"| `{n}` | {e.get('launches', '-')} | {fmt('avg_us')} | {fmt('mfma_busy_frac', 1.0, 3)} | {fmt('effective_clock_ghz', 1.0, 2)} | "
                    f"{fmt('read_bytes_avg', 1e-6)} | {fmt('write_bytes_avg', 1e-6)} | {fmt('hbm_GBps', 1.0, 0)} |\n")
        f.write("\n(the x2 on FETCH_SIZE is the gfx950 correction for 16-B-per-lane coalesced streams, MI355X_MICROARCH.md section HBM; "
                "kernels that read with narrower accesses are over-stated by it, Infinity-Cache hits are counted as fetches)\n")
print(open(os.path.join(P, f"{tag}_summary.md")).read())
