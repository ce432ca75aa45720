#!/bin/bash
# round 3, lab g: batch-split forward (two half-batch forwards side by side) against the plain forward
set -o pipefail
out=gpurun_out/r03lab_g; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "batch_split or test_forward" > $out/pytest.log 2>&1 || { tail -30 $out/pytest.log; exit 1; }
tail -1 $out/pytest.log
for w in bm403 dc237; do
  timeout -k 10 200 python tools/fwd_time.py --workload $w --tag "plain" >> $out/times.log 2>&1 || exit 1
  timeout -k 10 200 python tools/fwd_time.py --workload $w --split 2 --tag "split 2" >> $out/times.log 2>&1 || exit 1
  timeout -k 10 200 python tools/fwd_time.py --workload $w --cache-prepared --tag "plain, prepared cached" >> $out/times.log 2>&1 || exit 1
  timeout -k 10 200 python tools/fwd_time.py --workload $w --cache-prepared --split 2 --tag "split 2, prepared cached" >> $out/times.log 2>&1 || exit 1
done
grep -v amdgpu.ids $out/times.log
