#!/bin/bash
# usage: pmc_attn.sh "<counters...>"  — per-launch averages for the attention kernels of tools/exp/one_attn.py (CCEDIT_ATTN_PP selects the kernel)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pm1
rocprofv3 --pmc $1 --kernel-trace --output-format csv -d /tmp/pm1 -- python $GRAFT_REPO_ROOT/tools/exp/one_attn.py > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob("/tmp/pm1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "attn" in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k, (n, v) in sorted(acc.items()):
    print(f"  {k:40s} avg/launch {v/n:16.1f}  (n={n})")
PY
