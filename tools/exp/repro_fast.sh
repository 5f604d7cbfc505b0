run() { TAG="$1" env CCEDIT_OVERLAP_CONTROLNET=1 $1 REPS=6 timeout 600 python tools/exp/repro_fast.py 2>&1 | grep -v amdgpu | tail -1 | cut -c1-160; }
run "HSA_DISABLE_CACHE=1"
run "GPU_MAX_HW_QUEUES=1"
run "HIP_FORCE_DEV_KERNARG=0"
run "AMD_DIRECT_DISPATCH=0"
run "X=4"
