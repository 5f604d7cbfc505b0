#!/bin/bash
# usage: tools/pmc.sh <case> [args]  -- two PMC passes (SQ, then cache) over tools/mb_one.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pmc1 /tmp/pmc2
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pmc1 -- python $R/tools/mb_one.py "$@" > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT TCC_MISS TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d /tmp/pmc2 -- python $R/tools/mb_one.py "$@" > /dev/null 2>&1
python $R/tools/pmc_summary.py /tmp/pmc1 | grep -A12 -E "tap_gemm|attn_kernel"
python $R/tools/pmc_summary.py /tmp/pmc2 | grep -A12 -E "tap_gemm|attn_kernel"
