#!/bin/bash
# a variant of the kernel library with extra -D flags for ONE source: build_variant_of.sh <source.hip> <out.so> <flags...>
# (all other objects are the ones of the last regular build: run ccedit_amd/csrc/build.py first)
set -e
cd "$(dirname "$0")/../../ccedit_amd/csrc"
src=$1; out=$2; shift; shift
base=${src%.hip}; tmpo=$(mktemp /tmp/${base}_XXXX.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -x hip -c $src -o $tmpo
objs=""
for f in *.o; do [ "$f" != ${base}.o ] && objs="$objs $f"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $tmpo -o "$out"
