#!/bin/bash
# PMC passes over two training steps with every kernel alone on one stream (train_step.py ... serial), counters only:
#   1. MFMA utilisation of every kernel of the step (forward that keeps activations + backward)
#   2. LDS bank conflicts against LDS-active cycles (the GEMM staging the verdict asked about)
# usage: pmc_train_run.sh <tag>  -> gpurun_out/pmc_train_<tag>/{mfma,lds}/..., train_pmc_kernels.json
set -o pipefail
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_train_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/mfma -- python3 $R/tools/train_step.py bm403 2 serial > $OUT/mfma.log 2>&1
echo "[pmc] mfma rc=$?" | tee -a $OUT/progress.log
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/lds -- python3 $R/tools/train_step.py bm403 2 serial > $OUT/lds.log 2>&1
echo "[pmc] lds rc=$?" | tee -a $OUT/progress.log
cd $R
M=$(find $OUT/mfma -name "*counter_collection.csv" | head -1)
L=$(find $OUT/lds -name "*counter_collection.csv" | head -1)
python3 tools/summarise_train_pmc.py "$M" "$L" $OUT/train_pmc_kernels.json $TAG | tee $OUT/train_pmc_kernels.txt
