#!/bin/bash
# round 3, lab p: the head in two launches (all but the last two steps beside the top layer's tail) - parity, then time
set -o pipefail
out=gpurun_out/r03lab_p; mkdir -p $out; rm -f $out/times.log
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_model_gpu.py -m gpu -q -x > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $out/pytest.log
for rep in 1 2 3; do
  for w in bm403 dc237; do
  timeout -k 10 200 python tools/fwd_time.py --workload $w --kernels --tag "head split rep $rep" >> $out/times.log 2>&1 || exit 1
  done
done
grep -v amdgpu.ids $out/times.log | sort
