#!/bin/bash
# round 4, lab s2: finer sweep of GPU_MAX_HW_QUEUES (see lab_r04s.sh) - where is the best setting, where does it start to hurt
set -o pipefail
out=gpurun_out/r04lab_s2; mkdir -p $out; rm -f $out/times.log
for q in 5 6 8 10 12; do
  export GPU_MAX_HW_QUEUES=$q
  timeout -k 10 200 python tools/fwd_time.py --workload bm403 --train --tag "queues=$q" >> $out/times.log 2>&1 || exit 1
  bash tools/rehearse_rccl_1rank.sh > $out/rccl_$q.log 2>&1 || exit 1
  echo "queues=$q  RCCL 1-rank bench: $(tail -2 $out/rccl_$q.log | head -1)" >> $out/times.log
  echo "queues=$q  RCCL 1-rank train: $(tail -1 $out/rccl_$q.log)" >> $out/times.log
done
grep -v amdgpu.ids $out/times.log | cut -c1-220
