#!/bin/bash
# usage (on the GPU box): tools/pmc_counters.sh <kernel-name-regex> <mb_one case> [args]
# Two counter passes (SQ issue / matrix pipe, then VALU activity / LDS / clock) over tools/mb_one.py, --pmc with --kernel-trace only.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
PAT=$1; shift
rm -rf /tmp/pmcA /tmp/pmcB
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace --output-format csv -d /tmp/pmcA -- python $R/tools/mb_one.py "$@" > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAVES SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmcB -- python $R/tools/mb_one.py "$@" > /dev/null 2>&1
python $R/tools/pmc_summary.py /tmp/pmcA | grep -A9 -E "$PAT"
python $R/tools/pmc_summary.py /tmp/pmcB | grep -A9 -E "$PAT"
python - <<PY
import csv, glob
for d in ("/tmp/pmcA",):
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        import collections, re
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:70]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        for k, v in acc.items():
            if re.search(r"$PAT", k):
                print(f"{k}: {len(v)} dispatches, mean {sum(v)/len(v):.1f} us (under PMC)")
PY
