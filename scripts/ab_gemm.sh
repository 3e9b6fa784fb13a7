#!/bin/bash
# A/B of libwf3d builds / env switches on the split GEMM shapes, same box, one process per arm, arms repeated in
# alternation (scripts/bench_gemm.py prints medians over interleaved rounds inside a process).
#   gpurun -- 'bash scripts/ab_gemm.sh "old:WF3D_LIB=wireframe-3d-prediction_amd/libwf3d_old.so" "new:" "d8:WF3D_DEPHASE_TIMING_ONLY=8"'
R=${GRAFT_REPO_ROOT:-$(pwd)}
export ONLY=${ONLY:-SPLIT} ROUNDS=${ROUNDS:-7}
for rep in 1 2; do
  for arm in "$@"; do
    name=${arm%%:*}; envs=${arm#*:}
    echo "== $name (rep $rep) $envs"
    ( for kv in $envs; do export "$kv"; done; python3 $R/scripts/bench_gemm.py | grep -E "SPLIT (NT|dgrad|wgrad\(TN)" )
  done
done
