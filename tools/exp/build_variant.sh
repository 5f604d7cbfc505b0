#!/bin/bash
# build a variant of the kernel library with extra -D flags for ff320.hip: build_variant.sh <out.so> <flags...>
# (all other objects are the ones of the last regular build: run ccedit_amd/csrc/build.py first)
set -e
cd "$(dirname "$0")/../../ccedit_amd/csrc"
out=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize "$@" -x hip -c ff320.hip -o /tmp/ff320_var.o
objs=""
for f in *.o; do [ "$f" != ff320.o ] && objs="$objs $f"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/ff320_var.o -o "$out"
