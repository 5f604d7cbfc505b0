#!/bin/bash
# ablations of the fused feed-forward (fused timings only) + SQ counters of the real kernel
cd $GRAFT_REPO_ROOT
for a in 0 1 2 3 4; do echo "ABL=$a"; CCEDIT_FF320_ABL=$a timeout 120 python tools/exp/ff320_time.py 2>&1 | grep "^fused" | tail -1; done
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INST_LEVEL_VMEM" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SALU SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_VALU_MFMA_COEXEC_CYCLES"; do
  rm -rf /tmp/pmf
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmf -- python $GRAFT_REPO_ROOT/tools/exp/ff320_time.py > /dev/null 2>&1
  python $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/pmf | grep -A10 "ff320"
done
