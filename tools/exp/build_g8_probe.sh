#!/bin/bash
# libccedit_hip with gemm8p.hip compiled -DG8_PROBE (cycle stamps per workgroup into CcGemmDesc.gn_stats): build_g8_probe.sh <out.so>
set -e
cd "$(dirname "$0")/../../ccedit_amd/csrc"
out=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DG8_PROBE "$@" -x hip -c gemm8p.hip -o /tmp/gemm8p_probe.o
objs=""
for f in gemm convhalo smallconv lin320 ff320 norm attention attnspatial attnshort attntext elementwise core; do objs="$objs $f.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/gemm8p_probe.o -o "$out"
