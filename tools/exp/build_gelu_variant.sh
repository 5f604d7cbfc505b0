#!/bin/bash
# Variant of the kernel library with the GELU formula of rounds 1-5 (A&S 7.1.28) in the three files that evaluate it, every other object
# from the last regular build (run ccedit_amd/csrc/build.py first).  Output: build_var/libccedit_gelu28.so  (use with CCEDIT_HIP_LIB=...)
set -e
cd "$(dirname "$0")/../../ccedit_amd/csrc"
mkdir -p ../../build_var
objs=""
for f in *.o; do
  case $f in ff320.o|gemm8p.o|gemm.o|convhalo.o) ;; *) objs="$objs $f";; esac
done
for f in ff320 gemm8p gemm convhalo; do
  extra=""; [ $f = ff320 ] && extra="-fno-slp-vectorize"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $extra -DCCEDIT_GELU_AS71_28 -x hip -c $f.hip -o /tmp/${f}_g28.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/ff320_g28.o /tmp/gemm8p_g28.o /tmp/gemm_g28.o /tmp/convhalo_g28.o -o ../../build_var/libccedit_gelu28.so
ls -la ../../build_var/libccedit_gelu28.so
