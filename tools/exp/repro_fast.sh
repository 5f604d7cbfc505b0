# the evaluation N times in one process, bitwise: default (ControlNet on a side stream), TVI2V is covered by tests/test_fullsize_gpu.py
TAG=default REPS=${REPS:-30} timeout 900 python tools/exp/repro_fast.py 2>&1 | grep -v amdgpu | tail -1 | cut -c1-300
TAG=split CCEDIT_SPLIT_CFG=1 REPS=${REPS:-30} timeout 900 python tools/exp/repro_fast.py 2>&1 | grep -v amdgpu | tail -1 | cut -c1-300
