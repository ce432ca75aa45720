#!/bin/bash
# MFMA-utilisation PMC pass over two training steps (tools/train_step.py); counters only, no tracing domains.
set -o pipefail
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_mfma_train_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT -- python3 $R/tools/train_step.py bm403 2 > $OUT/run.log 2>&1
echo "[pmc] rc=$?" | tee -a $OUT/progress.log
find $OUT -name "*counter_collection.csv"
