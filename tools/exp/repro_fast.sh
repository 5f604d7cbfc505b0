run() { TAG="$1" env $1 REPS=6 timeout 600 python tools/exp/repro_fast.py 2>&1 | grep -v amdgpu | tail -1 | cut -c1-220; }
run "PYTORCH_NO_CUDA_MEMORY_CACHING=1"
run "PYTORCH_NO_CUDA_MEMORY_CACHING=1 CCEDIT_G8=0"
run "AMD_SERIALIZE_KERNEL=3"
run "X=3"
