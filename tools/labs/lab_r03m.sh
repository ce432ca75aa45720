#!/bin/bash
# round 3, lab m: k_gate16 alone at 96 registers (5 waves per SIMD: one gate workgroup fits beside five k_mix workgroups)
set -o pipefail
out=gpurun_out/r03lab_m; mkdir -p $out
L=multistgraph_amd/lib
for rep in 1 2; do
for v in "" gate96; do
  if [ -z "$v" ]; then lib=$L/libmatgcn.so; else lib=$L/libmatgcn_$v.so; fi
  MATGCN_LIB=$lib timeout -k 10 200 python tools/fwd_time.py --workload bm403 --kernels --tag "${v:-base} rep $rep" >> $out/times.log 2>&1 || exit 1
done
done
grep -v amdgpu.ids $out/times.log | sort
