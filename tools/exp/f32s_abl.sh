#!/bin/bash
# Where an iteration of the six-product fp32 kernels goes (f32s_gemm_kernel: 128 -> 128 at 512 x 768; f32p_gemm_kernel: 512 -> 512 at
# 128 x 192): probe builds of f32vae.hip with one stage removed each (-DF32S_ABL=n, see the file), built and timed on the GPU box.
#   0 as shipped, 1 no split arithmetic, 2 no global loads in the loop, 3 no split + LDS stores in the loop, 4 no MFMAs, 5 (f32p) no workgroup barriers
cd "$(dirname "$0")/../.."
python ccedit_amd/csrc/build.py > /dev/null 2>&1          # the other objects of the library
for v in 0 1 2 3 4 5; do
  bash tools/exp/build_variant_of.sh f32vae.hip /tmp/libabl$v.so -DF32S_ABL=$v > /dev/null 2>&1 &
done
wait
for v in 0 1 2 3 4 5; do
  CCEDIT_HIP_LIB=/tmp/libabl$v.so python tools/exp/f32s_one.py "abl $v"
done
