#!/bin/bash
# Builds libwf3d_<tag>.so variants of one source with extra -D flags, for A/B timing in one process
# tree on the GPU box:  scripts/build_variants.sh gemm_split.hip stamp "-DWF3D_STAMP=1" ...
# Select with WF3D_LIB=wireframe-3d-prediction_amd/libwf3d_<tag>.so.
set -e
cd "$(dirname "$0")/../wireframe-3d-prediction_amd/csrc"
make -s -j4
src=$1; shift
base=${src%.*}
while [ $# -gt 1 ]; do
  tag=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $flags -c $src -o build/${base}_$tag.o
  objs=""
  for o in capi gemm rowops pool attn edge split gemm_split attn_mfma loss skinny optim cloud evalpost; do
    if [ "$o" != "$base" ]; then objs="$objs build/$o.o"; fi
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libwf3d_$tag.so $objs build/${base}_$tag.o
  echo built libwf3d_$tag.so "($flags)"
done
