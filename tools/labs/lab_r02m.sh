#!/bin/bash
# backward parity subset, the step timing, and the isolated (serial-mode) kernel durations of the chain
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r02m
python -m pytest tests/test_backward_gpu.py -x -q -m gpu -k "bm_small or plugin or series or h0 or static or synthetic or golden or reference or gemm" > $R/gpurun_out/r02m/pytest.log 2>&1
echo "pytest rc=$?"; tail -2 $R/gpurun_out/r02m/pytest.log
python3 $R/tools/host_enqueue_time.py bm403 6 2>&1 | tail -3
export TMPDIR=/tmp
cd /tmp
OUT=$R/gpurun_out/r02m/serial
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/train_step.py bm403 3 serial > $OUT/steps.log 2> $OUT/err.log
tail -2 $OUT/steps.log
