#!/bin/bash
# round 3, lab r: training step, base build against the current one - wavefront (20 steps) and serial schedule (sum of the kernels)
set -o pipefail
out=gpurun_out/r03lab_r; mkdir -p $out; rm -f $out/times.log
L=multistgraph_amd/lib
export TRAIN_STEPS=20
for rep in 1 2; do
for v in base ""; do
  if [ -z "$v" ]; then lib=$L/libmatgcn.so; else lib=$L/libmatgcn_$v.so; fi
  MATGCN_LIB=$lib timeout -k 10 300 python tools/fwd_time.py --workload bm403 --train --iters 20 --tag "${v:-new} wavefront rep $rep" >> $out/times.log 2>&1 || exit 1
  MATGCN_LIB=$lib timeout -k 10 300 python tools/fwd_time.py --workload bm403 --train --serial --iters 20 --tag "${v:-new} serial rep $rep" >> $out/times.log 2>&1 || exit 1
done
done
grep -v amdgpu.ids $out/times.log | sort
