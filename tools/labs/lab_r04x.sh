#!/bin/bash
# round 4, lab x: 64 x 32 mix tiles (k_mix_c32) at B = 64 too on the small graph (DC 237: 768 workgroups of k_mix = 3 per CU)
set -o pipefail
out=gpurun_out/r04lab_x; mkdir -p $out; rm -f $out/times.log
L=multistgraph_amd/lib
for rep in 1 2; do
for v in c32all ""; do
  if [ -z "$v" ]; then lib=$L/libmatgcn.so; else lib=$L/libmatgcn_$v.so; fi
  MATGCN_LIB=$lib timeout -k 10 200 python tools/fwd_time.py --workload dc237 --tag "${v:-new} rep $rep" >> $out/times.log 2>&1 || exit 1
  MATGCN_LIB=$lib timeout -k 10 200 python tools/fwd_time.py --workload dc237 --kernels --tag "${v:-new} rep $rep" >> $out/times.log 2>&1 || exit 1
  MATGCN_LIB=$lib timeout -k 10 200 python tools/fwd_time.py --workload bm403 --tag "${v:-new} rep $rep" >> $out/times.log 2>&1 || exit 1
done
done
grep -v amdgpu.ids $out/times.log | sort | cut -c1-250
