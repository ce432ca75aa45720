#!/bin/bash
# round 3, lab l: X_CHUNK_STEPS = 2 as the default: whole GPU suite + forward / training-step time
set -o pipefail
out=gpurun_out/r03lab_l; mkdir -p $out
timeout -k 10 1100 python -m pytest tests -m gpu -q > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
for w in bm403 dc237; do
  timeout -k 10 200 python tools/fwd_time.py --workload $w --train --tag "xchunk2 default" >> $out/times.log 2>&1 || exit 1
done
grep -v amdgpu.ids $out/times.log
