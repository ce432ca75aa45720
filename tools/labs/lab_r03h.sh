#!/bin/bash
# round 3, lab h: k_update16 requests the residual cell's x_t rows at kernel start (stored to LDS before the last chunk)
set -o pipefail
out=gpurun_out/r03lab_h; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_model_gpu.py tests/test_hidden_pad.py -m gpu -x -q > $out/pytest.log 2>&1 || { tail -30 $out/pytest.log; exit 1; }
tail -1 $out/pytest.log
timeout -k 10 600 python -m pytest tests/test_backward_gpu.py -m gpu -x -q -k "reference_autograd or plugin_training" > $out/pytest_bwd.log 2>&1 || { tail -30 $out/pytest_bwd.log; exit 1; }
tail -1 $out/pytest_bwd.log
for w in bm403 dc237; do
  timeout -k 10 200 python tools/fwd_time.py --workload $w --kernels --train --tag "x_t early" >> $out/times.log 2>&1 || exit 1
done
grep -v amdgpu.ids $out/times.log
