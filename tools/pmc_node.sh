#!/bin/bash
# PMC passes over the node-kernel lab (counters only; no tracing domains).  usage: pmc_node.sh g|u
export TMPDIR=/tmp
cd /tmp
R=$GRAFT_REPO_ROOT
K=${1:-g}
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/pmcn1 -- $R/tools/nodelab $K > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM --output-format csv -d $R/gpurun_out/pmcn2 -- $R/tools/nodelab $K > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/pmcn3 -- $R/tools/nodelab $K > /dev/null 2>&1
find $R/gpurun_out/pmcn1 $R/gpurun_out/pmcn2 $R/gpurun_out/pmcn3 -name "*counter_collection.csv" | head
