#!/bin/bash
# the row-shard network cases one by one, each under its own timeout, output appended to gpurun_out/shard_cases.log as it comes
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/shard_cases.log
for k in "1-True-rows-rccl" "1-False-rows-gather-rccl" "2-False-rows]" "2-True-rows-gather" "4-False-rows"; do
  echo "=== $k $(date +%T)" >> gpurun_out/shard_cases.log
  timeout ${CASE_TIMEOUT:-330} python -m pytest tests/test_frame_shard_gpu.py -x -q -s -k "$k" >> gpurun_out/shard_cases.log 2>&1
  echo "=== rc $? $(date +%T)" >> gpurun_out/shard_cases.log
done
grep -E "^===|passed|failed|rel|Error|error" gpurun_out/shard_cases.log | tail -40
