#!/bin/bash
# round 4, lab af: fewer hardware queues than the default 4 (no process group)
set -o pipefail
out=gpurun_out/r04lab_af; mkdir -p $out; rm -f $out/times.log
for q in 2 3 4; do
  export GPU_MAX_HW_QUEUES=$q
  timeout -k 10 200 python tools/fwd_time.py --workload bm403 --train --tag "queues=$q" >> $out/times.log 2>&1 || exit 1
  timeout -k 10 200 python tools/fwd_time.py --workload bm403 --batch 16 --train --tag "queues=$q B=16" >> $out/times.log 2>&1 || exit 1
done
grep -v amdgpu.ids $out/times.log | cut -c1-220
