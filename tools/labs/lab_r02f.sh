#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r02f
python -m pytest tests/test_backward_gpu.py -x -q -m gpu -k "bucket or plugin_training" > gpurun_out/r02f/pytest.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/r02f/pytest.log
python bench.py > gpurun_out/r02f/bench.json 2> gpurun_out/r02f/bench.err
echo "bench rc=$?"; tail -c 600 gpurun_out/r02f/bench.err
bash tools/profile_r02.sh r02v13 > gpurun_out/r02f/profile.log 2>&1
echo "profile rc=$?"; tail -12 gpurun_out/r02f/profile.log
