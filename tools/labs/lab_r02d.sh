#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r02d
python -m pytest tests/test_hip_parity.py tests/test_model_gpu.py tests/test_windows.py -x -q -m gpu -k "not synth4096" > gpurun_out/r02d/pytest.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/r02d/pytest.log
python -m pytest tests/test_backward_gpu.py -x -q -m gpu > gpurun_out/r02d/pytest_bwd.log 2>&1
echo "pytest bwd rc=$?"; tail -6 gpurun_out/r02d/pytest_bwd.log
