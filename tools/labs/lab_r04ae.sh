#!/bin/bash
# round 4, lab ae: saved activations through write-through + NON-TEMPORAL stores (savent: aux sc1 | nt) against write-through (new)
set -o pipefail
out=gpurun_out/r04lab_ae; mkdir -p $out; rm -f $out/times.log
L=multistgraph_amd/lib
for rep in 1 2 3; do
for v in savent ""; do
  if [ -z "$v" ]; then lib=$L/libmatgcn.so; else lib=$L/libmatgcn_$v.so; fi
  MATGCN_LIB=$lib timeout -k 10 200 python tools/fwd_time.py --workload bm403 --train --tag "${v:-new} rep $rep" >> $out/times.log 2>&1 || exit 1
  MATGCN_LIB=$lib timeout -k 10 200 python tools/fwd_time.py --workload dc237 --train --tag "${v:-new} rep $rep" >> $out/times.log 2>&1 || exit 1
done
done
grep -v amdgpu.ids $out/times.log | sort | cut -c1-220
