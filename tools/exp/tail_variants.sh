#!/bin/bash
# tail_time.py for every build_var/tail_*.so (probe builds of ff320.hip with different -DTAIL_* knobs)
cd $GRAFT_REPO_ROOT
for so in build_var/tail_*.so; do
  echo "== $so"
  CCEDIT_HIP_LIB=$GRAFT_REPO_ROOT/$so python tools/exp/tail_time.py 2>&1 | grep -E "block tail|to_out fused|three|alone" | tail -4
done
