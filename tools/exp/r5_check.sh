#!/bin/bash
# round-5 GPU check bundle; everything is appended to gpurun_out/r5_check.log as it comes
cd $GRAFT_REPO_ROOT
L=gpurun_out/r5_check.log
mkdir -p gpurun_out; : > $L
run() { echo "=== $1 $(date +%T)" >> $L; shift; timeout "$@" >> $L 2>&1; echo "=== rc $? $(date +%T)" >> $L; }
run "shared prefix network test" 400 python -m pytest tests/test_network_gpu.py -x -q -s -k "shared_cfg or graph"
run "bench default (no clip/cpu)" 400 python bench.py --steps 20 --warmup 3 --no-clip --no-cpu-baseline --no-tvi2v --breakdown
cp $L /tmp/x; python bench.py --steps 20 --warmup 3 --no-clip --no-cpu-baseline --no-tvi2v > gpurun_out/r5_d.json 2>/dev/null
CCEDIT_POLICY=share_cfg_prefix=0 python bench.py --steps 20 --warmup 3 --no-clip --no-cpu-baseline --no-tvi2v > gpurun_out/r5_d_noshare.json 2>/dev/null
run "rows-rccl cross (teardown)" 200 python -m pytest tests/test_frame_shard_gpu.py -x -q -s -k "1-True-rows-rccl"
SHARD_TEST_DUMP_S=150 run "world 4 rows" 260 python -m pytest tests/test_frame_shard_gpu.py -x -q -s -k "4-False-rows"
run "low rows sweep" 300 python tools/exp/low_rows.py
grep -E "^===|passed|failed" $L | tail -30
