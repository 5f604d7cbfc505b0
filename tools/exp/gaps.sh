cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pf_gaps
CCEDIT_SPLIT_CFG=0 CCEDIT_OVERLAP_CONTROLNET=0 rocprofv3 --kernel-trace --output-format csv -d /tmp/pf_gaps -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-clip --no-tvi2v --no-c4 2>/dev/null | tail -1 | cut -c1-200
python $GRAFT_REPO_ROOT/tools/exp/gaps.py /tmp/pf_gaps
