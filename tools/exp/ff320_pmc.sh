#!/bin/bash
# SQ counters of the fused feed-forward kernel; usage: ff320_pmc.sh [ABL]
cd /tmp && export TMPDIR=/tmp
export CCEDIT_FF320_ABL=${1:-0}
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" "GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_COEXEC_CYCLES"; do
  rm -rf /tmp/pmf
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmf -- python $GRAFT_REPO_ROOT/tools/exp/ff320_time.py > /dev/null 2>&1
  python $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/pmf | grep -A9 "ff320"
done
