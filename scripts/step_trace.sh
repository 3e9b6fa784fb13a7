#!/bin/bash
# Kernel-by-kernel timeline of ONE benchmark step (rocprofv3 --kernel-trace over bench.py): name, start offset, duration,
# gap to the previous kernel.   gpurun -- 'bash scripts/step_trace.sh r03 cfg2'   ->  gpurun_out/steptrace_<tag>.txt
TAG=${1:?tag}; CFG=${2:-cfg2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/st_$TAG -- python3 $R/bench.py --config $CFG --no-cpu-baseline --no-power-probe --steps 3 --warmup 2 > $R/gpurun_out/st_$TAG.log 2>&1
python3 - "$R/gpurun_out/st_$TAG" > $R/gpurun_out/steptrace_$TAG.txt <<'PY'
import csv, glob, sys
d = sys.argv[1]
rows = []
for f in glob.glob(d + "/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:64]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n))
for f in glob.glob(d + "/**/*_memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "MEMCPY " + r.get("Direction", "") + " " + r.get("Bytes", "")))
rows.sort()
# one step = from a first_layer_kernel launch to the next one; take the last complete step
starts = [i for i, r in enumerate(rows) if r[2].startswith("first_layer_kernel")]
a, b = starts[-2], starts[-1]
t0 = rows[a][0]
prev_end = rows[a][0]
print(f"# step of {b - a} launches / copies, {(rows[b][0] - t0) / 1e6:.3f} ms")
tot = 0
for s, e, n in rows[a:b]:
    print(f"{(s - t0) / 1e3:10.1f} us  dur {(e - s) / 1e3:8.1f}  gap {(s - prev_end) / 1e3:7.1f}  {n}")
    prev_end = max(prev_end, e)
    tot += e - s
print(f"# sum of durations {tot / 1e6:.3f} ms")
PY
head -1 $R/gpurun_out/steptrace_$TAG.txt
