#!/bin/bash
# PMC pass over the mix lab (counters only; no tracing domains)
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/counters.txt 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc1 -- $GRAFT_REPO_ROOT/tools/mixlab > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc2 -- $GRAFT_REPO_ROOT/tools/mixlab > /dev/null 2>&1
ls -R $GRAFT_REPO_ROOT/gpurun_out/pmc1 | head
