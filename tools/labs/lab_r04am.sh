#!/bin/bash
# round 4, lab am: FREE-RUNNING training loops in the own-pool mode at 4 and 8 queues (the default streams were bimodal at 8: lab ab)
set -o pipefail
out=gpurun_out/r04lab_am; mkdir -p $out; rm -f $out/times.log
export MATGCN_POOL=1
for rep in 1 2 3; do
for q in 4 8; do
  export GPU_MAX_HW_QUEUES=$q
  timeout -k 10 200 python tools/train_loop_wall.py bm403 64 >> $out/times.log 2>&1 || exit 1
  timeout -k 10 200 python tools/train_loop_wall.py bm403 16 >> $out/times.log 2>&1 || exit 1
  timeout -k 10 200 python tools/train_loop_wall.py dc237 16 >> $out/times.log 2>&1 || exit 1
done
done
grep -v amdgpu.ids $out/times.log | sort | cut -c1-220
