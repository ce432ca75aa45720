#!/bin/bash
# round 4, lab an: FREE-RUNNING training loops inside a one-rank RCCL process group: default streams at 4 and 8 queues against the own pool at 4
set -o pipefail
out=gpurun_out/r04lab_an; mkdir -p $out; rm -f $out/times.log
export MATGCN_PG=1
for rep in 1 2; do
for cfg in "0 4" "0 8" "1 4"; do
  set -- $cfg
  export MATGCN_POOL=$1 GPU_MAX_HW_QUEUES=$2
  timeout -k 10 200 python tools/train_loop_wall.py bm403 64 >> $out/times.log 2>&1 || exit 1
  timeout -k 10 200 python tools/train_loop_wall.py bm403 16 >> $out/times.log 2>&1 || exit 1
done
done
grep -v amdgpu.ids $out/times.log | grep "free-running" | sort | cut -c1-230
