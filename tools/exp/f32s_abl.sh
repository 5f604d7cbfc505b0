#!/bin/bash
# where an iteration of f32s_gemm_kernel goes: probe builds of f32vae.hip with one stage removed each (run on the GPU box)
cd "$(dirname "$0")/../.."
for v in 0 1 2 3 4 5; do
  CCEDIT_HIP_LIB=$PWD/tools/exp/_abl/libabl$v.so python tools/exp/f32s_one.py "abl $v"
done
