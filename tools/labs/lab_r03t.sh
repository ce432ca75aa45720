#!/bin/bash
# round 3, lab t: stream priorities for the upper layers' chain / x-part streams of the forward (the dependent chain first)
set -o pipefail
out=gpurun_out/r03lab_t; mkdir -p $out; rm -f $out/times.log
L=multistgraph_amd/lib/libmatgcn_prio.so
for rep in 1 2; do
for cfg in "none none" "-1 none" "-1 -1" "none -1" "1 none"; do
  set -- $cfg
  unset MATGCN_CHAIN_PRIORITY MATGCN_XPART_PRIORITY
  [ "$1" != none ] && export MATGCN_CHAIN_PRIORITY=$1
  [ "$2" != none ] && export MATGCN_XPART_PRIORITY=$2
  for w in bm403 dc237; do
    MATGCN_LIB=$L timeout -k 10 200 python tools/fwd_time.py --workload $w --tag "chain prio $1 xpart prio $2 rep $rep" >> $out/times.log 2>&1 || exit 1
  done
done
done
grep -v amdgpu.ids $out/times.log | sort
