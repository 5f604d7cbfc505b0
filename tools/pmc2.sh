#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pmc3
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVES SQ_LEVEL_WAVES SQ_INSTS_SALU --kernel-trace --output-format csv -d /tmp/pmc3 -- python $R/tools/mb_one.py "$@" > /dev/null 2>&1
python $R/tools/pmc_summary.py /tmp/pmc3 | grep -A12 -E "tap_gemm|attn_kernel"
rm -rf /tmp/pmc4
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_THREAD_CYCLES_VALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_MISC --kernel-trace --output-format csv -d /tmp/pmc4 -- python $R/tools/mb_one.py "$@" > /dev/null 2>&1
python $R/tools/pmc_summary.py /tmp/pmc4 | grep -A12 -E "tap_gemm|attn_kernel"
