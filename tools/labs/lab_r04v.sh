#!/bin/bash
# round 4, lab v: GPU_MAX_HW_QUEUES 4 (the runtime's default) against 8 (the package's default since lab s) on the OTHER workloads,
# no process group: DC 237, the shipped batch size 16, N = 4096
set -o pipefail
out=gpurun_out/r04lab_v; mkdir -p $out; rm -f $out/times.log
for rep in 1 2; do
for q in 4 8; do
  export GPU_MAX_HW_QUEUES=$q
  timeout -k 10 200 python tools/fwd_time.py --workload dc237 --train --tag "queues=$q rep $rep" >> $out/times.log 2>&1 || exit 1
  timeout -k 10 200 python tools/fwd_time.py --workload bm403 --batch 16 --train --tag "queues=$q B=16 rep $rep" >> $out/times.log 2>&1 || exit 1
  timeout -k 10 200 python tools/fwd_time.py --workload dc237 --batch 16 --train --tag "queues=$q B=16 rep $rep" >> $out/times.log 2>&1 || exit 1
done
done
for q in 4 8; do
  export GPU_MAX_HW_QUEUES=$q
  timeout -k 10 300 python tools/fwd_time.py --workload synth4096 --iters 5 --train --tag "queues=$q" >> $out/times.log 2>&1 || exit 1
done
grep -v amdgpu.ids $out/times.log | sort | cut -c1-220
