#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r02e
python -m pytest tests -q -m gpu > gpurun_out/r02e/pytest.log 2>&1
echo "pytest rc=$?"; tail -8 gpurun_out/r02e/pytest.log
