#!/bin/bash
# calibrate SQ_VALU_MFMA_BUSY_CYCLES on a pure-MFMA kernel (mixlab2 reg16) and read it for the gate kernel
export TMPDIR=/tmp
cd /tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_cal_mix -- $R/tools/mixlab2 > /dev/null 2>&1
echo "[calib] mixlab2 done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_cal_gate -- $R/tools/nodelab g > /dev/null 2>&1
echo "[calib] nodelab done"
