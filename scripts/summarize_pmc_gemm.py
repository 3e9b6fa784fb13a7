#!/usr/bin/env python3
"""Per-kernel counter table of the split GEMMs from the passes scripts/pmc_gemm.sh left under gpurun_out/pmc_<tag>_*.

    python scripts/summarize_pmc_gemm.py r03a            -> profiles/<tag>_gemm_pmc.md / .json

Every figure is the mean over the launches of one kernel that ran longer than MIN_US (the encoder-shape launches; the
warm-up and the small helper kernels are left out).  Units (MI355X_MICROARCH.md, rocprofv3 PMC slots): SQ_WAVE_CYCLES /
SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD pipe;
SQ_LDS_IDX_ACTIVE / SQ_LDS_BANK_CONFLICT count LDS-array cycles per CU; GRBM_GUI_ACTIVE is summed over the 8 XCDs.
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03a"
MIN_US = float(os.environ.get("MIN_US", 250))
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
CUS, SIMDS = 256, 1024


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:60]


def load(kind):
    fs = sorted(glob.glob(os.path.join(G, f"pmc_{tag}_{kind}", "**", "*_counter_collection.csv"), recursive=True))
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    if not fs:
        return out
    for r in csv.DictReader(open(fs[-1])):
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        if dur < MIN_US:
            continue
        k = short(r["Kernel_Name"])
        out[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        out[k]["_dur_us"].append(dur)
    return out


def mean(v):
    return sum(v) / len(v) if v else None


passes = {k: load(k) for k in ("lds", "sq", "tcc", "tcp")}
kernels = sorted(set(passes["sq"]) | set(passes["lds"]))
res = {}
for k in kernels:
    e = {}
    for pname, p in passes.items():
        for c, v in p.get(k, {}).items():
            if c == "_dur_us":
                # every counter row of a dispatch repeats the duration: de-duplicate by taking the mean
                e[f"dur_us_{pname}"] = mean(v)
                continue
            e[c] = mean(v)
        if k in p:
            ncount = len([c for c in p[k] if c != "_dur_us"])
            e[f"launches_{pname}"] = len(p[k]["_dur_us"]) // max(ncount, 1)
    d = e.get("dur_us_sq")
    gui = e.get("GRBM_GUI_ACTIVE")
    if d and gui:
        clk = gui / 8.0 / (d * 1e3)                         # GHz
        e["clock_ghz"] = clk
        cyc = gui / 8.0                                      # shader cycles of the launch
        e["mfma_busy"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * SIMDS) if e.get("SQ_VALU_MFMA_BUSY_CYCLES") else None
        wc = e.get("SQ_WAVE_CYCLES")
        if wc:
            for c in ("SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"):
                if e.get(c) is not None:
                    e[c + "_per_wave_cycle"] = e[c] / wc
    dl = e.get("dur_us_lds")
    if dl and e.get("clock_ghz"):
        cyc = dl * 1e3 * e["clock_ghz"]
        if e.get("SQ_LDS_IDX_ACTIVE") is not None:
            e["lds_busy"] = e["SQ_LDS_IDX_ACTIVE"] / (cyc * CUS)
            e["lds_conflict_frac"] = e["SQ_LDS_BANK_CONFLICT"] / e["SQ_LDS_IDX_ACTIVE"] if e["SQ_LDS_IDX_ACTIVE"] else None
        wc = e.get("SQ_WAVE_CYCLES")
        if wc:
            for c in ("SQ_WAIT_INST_LDS", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM",
                      "SQ_LDS_DATA_FIFO_FULL", "SQ_LDS_CMD_FIFO_FULL"):
                if e.get(c) is not None:
                    e[c + "_per_wave_cycle"] = e[c] / wc
    if e.get("TCC_HIT_sum") is not None and e.get("TCC_MISS_sum") is not None:
        tot = e["TCC_HIT_sum"] + e["TCC_MISS_sum"]
        e["l2_hit_rate"] = e["TCC_HIT_sum"] / tot if tot else None
    res[k] = e

os.makedirs(P, exist_ok=True)
json.dump(res, open(os.path.join(P, f"{tag}_gemm_pmc.json"), "w"), indent=1, sort_keys=True)


def f(v, fmt="{:.3f}"):
    return "—" if v is None else fmt.format(v)


cols = [("clock_ghz", "clock GHz"), ("mfma_busy", "MFMA busy"), ("lds_busy", "LDS-array busy"), ("lds_conflict_frac", "bank-conflict share of LDS cycles"),
        ("SQ_WAIT_ANY_per_wave_cycle", "WAIT_ANY / wave cyc"), ("SQ_WAIT_INST_ANY_per_wave_cycle", "WAIT_INST_ANY / wave cyc"),
        ("SQ_WAIT_INST_LDS_per_wave_cycle", "WAIT_INST_LDS / wave cyc"), ("SQ_ACTIVE_INST_ANY_per_wave_cycle", "ACTIVE_INST_ANY / wave cyc"),
        ("SQ_ACTIVE_INST_LDS_per_wave_cycle", "ACTIVE_INST_LDS / wave cyc"), ("SQ_ACTIVE_INST_VMEM_per_wave_cycle", "ACTIVE_INST_VMEM / wave cyc"),
        ("SQ_LDS_DATA_FIFO_FULL_per_wave_cycle", "LDS_DATA_FIFO_FULL / wave cyc"), ("SQ_LDS_CMD_FIFO_FULL_per_wave_cycle", "LDS_CMD_FIFO_FULL / wave cyc"),
        ("l2_hit_rate", "L2 hit rate"), ("TCC_REQ_sum", "TCC_REQ per launch")]
with open(os.path.join(P, f"{tag}_gemm_pmc.md"), "w") as o:
    o.write(f"# Split-GEMM counters, tag {tag} (scripts/pmc_gemm.sh: bench_gemm.py ONLY=SPLIT, launches > {MIN_US:.0f} us)\n\n")
    o.write("| kernel | launches | mean us | " + " | ".join(c[1] for c in cols) + " |\n")
    o.write("|---|---|---|" + "---|" * len(cols) + "\n")
    for k, e in res.items():
        o.write(f"| `{k}` | {e.get('launches_sq', e.get('launches_lds', 0))} | {f(e.get('dur_us_sq'), '{:.0f}')} | " +
                " | ".join(f(e.get(c[0]), "{:.3g}") for c in cols) + " |\n")
    o.write("\nRaw means per launch: `" + f"{tag}_gemm_pmc.json" + "`.\n")
print(open(os.path.join(P, f"{tag}_gemm_pmc.md")).read())
