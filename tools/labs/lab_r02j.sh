#!/bin/bash
# quick check of a backward change: a subset of the backward parity tests, then the training-step timing of bench.py
set -o pipefail
mkdir -p gpurun_out/r02j
python -m pytest tests/test_backward_gpu.py -x -q -m gpu -k "bm_small or plugin or series or h0 or static or synthetic" > gpurun_out/r02j/pytest.log 2>&1
echo "pytest rc=$?"; tail -2 gpurun_out/r02j/pytest.log
python tools/host_enqueue_time.py bm403 5 2>&1 | tail -3
