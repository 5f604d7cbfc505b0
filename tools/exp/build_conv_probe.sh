#!/bin/bash
# Probe build of the kernel library: convhalo.hip with -DCONV_PROBE (per-workgroup time stamps), every other object from the last
# regular build (run ccedit_amd/csrc/build.py first).  Output: build_var/libccedit_probe.so  (use with CCEDIT_HIP_LIB=...)
set -e
cd "$(dirname "$0")/../../ccedit_amd/csrc"
mkdir -p ../../build_var
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DCONV_PROBE "$@" -x hip -c convhalo.hip -o /tmp/convhalo_probe.o
objs=""
for f in *.o; do [ "$f" != convhalo.o ] && objs="$objs $f"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/convhalo_probe.o -o ../../build_var/libccedit_probe.so
ls -la ../../build_var/libccedit_probe.so
