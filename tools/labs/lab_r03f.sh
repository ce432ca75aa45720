#!/bin/bash
# round 3, lab f: what is fixed in a k_mix launch?  K loop cut to 2 / 14 of 26 tiles, stores removed (results garbage)
set -o pipefail
out=gpurun_out/r03lab_f; mkdir -p $out
L=multistgraph_amd/lib
for v in "" mixnk2 mixnk2ns mixns mixnk14; do
  if [ -z "$v" ]; then lib=$L/libmatgcn.so; else lib=$L/libmatgcn_$v.so; fi
  MATGCN_LIB=$lib timeout -k 10 200 python tools/fwd_time.py --kernels --tag "${v:-base}" >> $out/times.log 2>&1 || exit 1
done
grep -v amdgpu.ids $out/times.log
