#!/bin/bash
# stall / LDS PMC passes over two serial-schedule forwards (counters only).  usage: pmc_mix.sh <tag>
export TMPDIR=/tmp
cd /tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_mix_${1:-r01}
mkdir -p $OUT
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $OUT/p1 -- python3 $R/tools/one_forward.py > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT/p2 -- python3 $R/tools/one_forward.py > $OUT/p2.log 2>&1
find $OUT -name "*counter_collection.csv"
