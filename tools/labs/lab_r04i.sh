#!/bin/bash
# round 4, lab i: k_mix with the wave priority rotating over the K-tiles (prio1: every 2 tiles, prio3: every 8) against the product
set -o pipefail
out=gpurun_out/r04lab_i; mkdir -p $out; rm -f $out/times.log
L=multistgraph_amd/lib
for rep in 1 2; do
for v in "" prio1 prio3; do
  if [ -z "$v" ]; then lib=$L/libmatgcn.so; else lib=$L/libmatgcn_$v.so; fi
  for w in bm403 dc237; do
  MATGCN_LIB=$lib timeout -k 10 200 python tools/fwd_time.py --workload $w --kernels --tag "${v:-base} rep $rep" >> $out/times.log 2>&1 || exit 1
  done
done
done
grep -v amdgpu.ids $out/times.log | sort | cut -c1-330
MATGCN_LIB=$L/libmatgcn_stamps_prio1.so timeout -k 10 200 python tools/labs/stamps_mix_r04.py > $out/stamps_mix_prio1.log 2>&1; grep -v amdgpu $out/stamps_mix_prio1.log | head -16
