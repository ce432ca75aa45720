#!/bin/bash
# round 3, lab e: staggered start of the k_mix workgroups that share a CU (phase-locked identical programs)
set -o pipefail
out=gpurun_out/r03lab_e; mkdir -p $out
L=multistgraph_amd/lib
for v in "" stag4 stag8 stag12 stag16; do
  if [ -z "$v" ]; then lib=$L/libmatgcn.so; else lib=$L/libmatgcn_$v.so; fi
  MATGCN_LIB=$lib timeout -k 10 200 python tools/fwd_time.py --kernels --tag "${v:-base}" >> $out/times.log 2>&1 || exit 1
done
grep -v amdgpu.ids $out/times.log
