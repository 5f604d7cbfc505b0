#!/bin/bash
# build a variant of the kernel library with extra -D flags for ONE source: build_variant2.sh <out.so> <source stem> <flags...>
set -e
cd "$(dirname "$0")/../../ccedit_amd/csrc"
out=$1; stem=$2; shift; shift
extra=""
[ "$stem" = ff320 ] && extra="-fno-slp-vectorize"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $extra "$@" -x hip -c $stem.hip -o /tmp/${stem}_var_$$.o
objs=""
for f in gemm convhalo smallconv lin320 ff320 norm attention attnshort elementwise core; do
  [ "$f" = "$stem" ] && objs="$objs /tmp/${stem}_var_$$.o" || objs="$objs $f.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o "$out"
