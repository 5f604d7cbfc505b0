#!/bin/bash
# HBM traffic of one bench run: two PMC passes (FETCH_SIZE, WRITE_SIZE) per MI355X_MICROARCH.md; kernel-trace only.
cd /tmp && export TMPDIR=/tmp
# batched, single-stream evaluation: the same launches as the HIP-event-profiled step of bench.py
export CCEDIT_SPLIT_CFG=0 CCEDIT_OVERLAP_CONTROLNET=0
R=$GRAFT_REPO_ROOT
rm -rf /tmp/tr1 /tmp/tr2
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/tr1 -- python $R/bench.py $PMC_BENCH_ARGS --steps 1 --warmup 0 --no-cpu-baseline --no-clip --no-tvi2v --no-profile-step > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/tr2 -- python $R/bench.py $PMC_BENCH_ARGS --steps 1 --warmup 0 --no-cpu-baseline --no-clip --no-tvi2v --no-profile-step > /dev/null 2>&1
python $R/bench.py $PMC_BENCH_ARGS --steps 1 --warmup 0 --no-cpu-baseline --no-clip --no-tvi2v --no-profile-step --dump-shapes /tmp/shapes.json > /dev/null 2>&1
python $R/tools/pmc_traffic.py /tmp/tr1 /tmp/tr2 /tmp/shapes.json
