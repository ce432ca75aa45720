#!/bin/bash
# round 3, lab i: lazy prepare (the weight streams of matgcn_prepare run beside the start of the forward)
set -o pipefail
out=gpurun_out/r03lab_i; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_model_gpu.py tests/test_hidden_pad.py tests/test_windows.py tests/test_dataset.py -m gpu -x -q > $out/pytest.log 2>&1 || { tail -30 $out/pytest.log; exit 1; }
tail -1 $out/pytest.log
timeout -k 10 900 python -m pytest tests/test_backward_gpu.py -m gpu -x -q > $out/pytest_bwd.log 2>&1 || { tail -30 $out/pytest_bwd.log; exit 1; }
tail -1 $out/pytest_bwd.log
for w in bm403 dc237; do
  MATGCN_LAZY_PREPARE=0 timeout -k 10 200 python tools/fwd_time.py --workload $w --train --tag "eager prepare" >> $out/times.log 2>&1 || exit 1
  timeout -k 10 200 python tools/fwd_time.py --workload $w --train --tag "lazy prepare" >> $out/times.log 2>&1 || exit 1
  timeout -k 10 200 python tools/fwd_time.py --workload $w --cache-prepared --tag "prepared cached" >> $out/times.log 2>&1 || exit 1
done
grep -v amdgpu.ids $out/times.log
