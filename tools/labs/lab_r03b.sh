#!/bin/bash
# round 3, lab b: do the chains gain from disjoint CU sets (MATGCN_CU_SPLIT)?  outputs under gpurun_out/r03lab_b/
set -o pipefail
out=gpurun_out/r03lab_b; mkdir -p $out
timeout -k 10 120 tools/labs/cumask_probe > $out/cumask_probe.log 2>&1
python tools/fwd_time.py --tag "default wavefront" > $out/times.log 2>&1 || exit 1
for a in 8 12 14 16 18 20 24; do
  for x in 0 1; do
    MATGCN_CU_SPLIT=$a MATGCN_CU_SPLIT_X=$x timeout -k 10 200 python tools/fwd_time.py --tag "split $a / $((32-a)) xmode $x" >> $out/times.log 2>&1 || exit 1
  done
done
python tools/fwd_time.py --tag "default wavefront (again)" >> $out/times.log 2>&1
MATGCN_CU_SPLIT=16 timeout -k 10 200 python tools/fwd_time.py --workload dc237 --tag "split 16/16" >> $out/times.log 2>&1
python tools/fwd_time.py --workload dc237 --tag "default" >> $out/times.log 2>&1
cat $out/cumask_probe.log $out/times.log
