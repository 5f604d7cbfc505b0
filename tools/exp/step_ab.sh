#!/bin/bash
# Same-box A/B of the whole step: bench.py (step only) under a list of policy strings / library variants, three rounds, alternating.
#   tools/exp/step_ab.sh out.txt "name|CCEDIT_POLICY value|library path or -" ...
out=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
: > $out
for rnd in 1 2 3; do
  for spec in "$@"; do
    IFS='|' read -r name pol lib <<< "$spec"
    if [ "$lib" = "-" ] || [ -z "$lib" ]; then unset CCEDIT_HIP_LIB; else export CCEDIT_HIP_LIB=$R/$lib; fi
    ms=$(CCEDIT_POLICY="$pol" python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-clip --no-tvi2v --no-profile-step 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "round $rnd  $name: $ms ms/step" | tee -a $out
  done
done
unset CCEDIT_HIP_LIB
