#!/bin/bash
# Kernel-only durations of the split GEMMs (rocprofv3 --kernel-trace --stats over scripts/bench_gemm.py) for several
# library builds, one profiled process each:  bash scripts/ktrace_gemm.sh name1:lib1.so name2:lib2.so ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
export ONLY=${ONLY:-SPLIT} ROUNDS=${ROUNDS:-5}
for arm in "$@"; do
  name=${arm%%:*}; lib=${arm#*:}
  export WF3D_LIB=$R/$lib
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt_$name -- python3 $R/scripts/bench_gemm.py > $R/gpurun_out/kt_$name.log 2>&1
  f=$(ls $R/gpurun_out/kt_$name/*/*_kernel_stats.csv | tail -1)
  echo "== $name"; python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:50]
    if "gemm_split" in n or "fill" in n.lower() or "memset" in n.lower():
        print(f"   {n:52s} calls {r['Calls']:>4s}  total {float(r['TotalDurationNs'])/1e6:8.3f} ms  avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
done
