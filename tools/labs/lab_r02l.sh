#!/bin/bash
# backward parity subset on the product build, then step timings
set -o pipefail
mkdir -p gpurun_out/r02l
python -m pytest tests/test_backward_gpu.py -x -q -m gpu -k "bm_small or plugin or series or h0 or static or synthetic or golden or reference" > gpurun_out/r02l/pytest.log 2>&1
echo "pytest rc=$?"; tail -2 gpurun_out/r02l/pytest.log
python tools/host_enqueue_time.py bm403 6 2>&1 | tail -3
