#!/bin/bash
# the whole GPU suite, file by file, output appended to gpurun_out/r5_fulltests.log as it comes
cd $GRAFT_REPO_ROOT
L=gpurun_out/r5_fulltests.log
mkdir -p gpurun_out; : > $L
for f in tests/test_ops_gpu.py tests/test_vae_f32_gpu.py tests/test_network_gpu.py tests/test_entrypoints_gpu.py tests/test_fullsize_gpu.py tests/test_frame_shard_gpu.py; do
  echo "=== $f $(date +%T)" >> $L
  timeout ${FILE_TIMEOUT:-1500} python -m pytest $f -q -m gpu --durations=8 2>&1 | grep -vE "amdgpu.ids|socket.cpp|Gloo|^$" | tail -40 >> $L
  echo "=== rc ${PIPESTATUS[0]} $(date +%T)" >> $L
done
grep -E "^===|passed|failed" $L
