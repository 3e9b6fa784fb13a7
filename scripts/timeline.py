#!/usr/bin/env python3
"""Per-launch timeline of ONE bench step from a rocprofv3 --kernel-trace CSV (gpurun_out/prof_<tag>_trace2/**/_kernel_trace.csv):
start offset, gap to the previous kernel's end, duration, kernel, grid.  `python scripts/timeline.py <csv> [step] [lo_us hi_us]`"""
import csv
import re
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"\(.*$", "", n).replace("void ", "")
    if "at::native" in n:
        n = "ATen:" + re.sub(r"<.*", "", n.split("at::native::")[1])[:40]
    return n[:64]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "point_valid" in r["Kernel_Name"]]
    step = int(sys.argv[2]) if len(sys.argv) > 2 else -2
    a = marks[step]
    b = marks[step + 1] if step + 1 < len(marks) and step != -1 else len(rows)
    lo = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
    hi = float(sys.argv[4]) if len(sys.argv) > 4 else 1e12
    t0 = int(rows[a]["Start_Timestamp"])
    prev = t0
    n = 0
    for r in rows[a:b]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        off = (s - t0) / 1e3
        if lo <= off <= hi:
            print(f"{off:9.1f} gap{(s - prev) / 1e3:7.1f} dur{(e - s) / 1e3:8.1f}  {short(r['Kernel_Name'])}  grid={r['Grid_Size_X']} wg={r['Workgroup_Size_X']}")
        prev = e
        n += 1
    print(f"# {n} launches, step span {(prev - t0) / 1e3:.1f} us")


if __name__ == "__main__":
    main()
