TAG=base python tools/exp/attn_text_perf.py 2>&1 | grep -v amdgpu
TAG=generic CCEDIT_ATTN_TEXT=0 python tools/exp/attn_text_perf.py 2>&1 | grep -v amdgpu
for v in 1 2 3 4; do TAG=probe$v CCEDIT_HIP_LIB=$PWD/build_var/libccedit_at_p$v.so python tools/exp/attn_text_perf.py 2>&1 | grep -v amdgpu; done
