#!/bin/bash
# round 4, lab al: more arrangements inside the library's own queue pool (MATGCN_POOL=1 makes tools/fwd_time.py switch the mode on):
# b0 = shipped (main 1, chain[1] 0, xpart[1] 1, aux 2, xcol 3, bchain 1); b1 xpart[1] on chain[1]'s queue; b2-b4 aux at 0 / 1 / 3;
# b5 xpart[1] and bchain on chain[1]'s queue
set -o pipefail
out=gpurun_out/r04lab_al; mkdir -p $out; rm -f $out/times.log
L=$GRAFT_REPO_ROOT/multistgraph_amd/lib
export MATGCN_POOL=1
for rep in 1 2; do
for v in b0 b1 b2 b3 b4 b5; do
  if [ "$v" = "b0" ]; then export MATGCN_LIB=$L/libmatgcn.so; else export MATGCN_LIB=$L/libmatgcn_$v.so; fi
  timeout -k 10 200 python tools/fwd_time.py --workload bm403 --train --tag "$v rep $rep" >> $out/times.log 2>&1 || exit 1
  timeout -k 10 200 python tools/fwd_time.py --workload bm403 --batch 16 --train --tag "$v B=16 rep $rep" >> $out/times.log 2>&1 || exit 1
done
done
unset MATGCN_POOL MATGCN_LIB
timeout -k 10 200 python tools/fwd_time.py --workload bm403 --train --tag "default mode" >> $out/times.log 2>&1
grep -v amdgpu.ids $out/times.log | sort | cut -c1-200
