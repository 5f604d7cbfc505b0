cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_network_gpu.py -x -q > gpurun_out/q80_net.log 2>&1; tail -3 gpurun_out/q80_net.log
timeout 900 python -m pytest tests/test_fullsize_gpu.py -x -q -k "input_blocks.4 or input_blocks.5 or input_blocks.7" > gpurun_out/q80_full.log 2>&1; tail -3 gpurun_out/q80_full.log
for i in 1 2; do timeout 600 python bench.py --steps 8 --warmup 3 --no-clip --no-cpu-baseline --no-tvi2v --no-c4 2>gpurun_out/q80_b.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
