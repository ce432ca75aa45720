#!/bin/bash
# round 4, lab t: creation order of the library's streams (= the order the runtime hands out hardware queues): the streams a
# two-layer model keeps busy together first and back to back (new) against chain/xpart of all layers first (oldorder)
set -o pipefail
out=gpurun_out/r04lab_t; mkdir -p $out; rm -f $out/times.log
L=$GRAFT_REPO_ROOT/multistgraph_amd/lib
for q in 4 8; do
for v in oldorder ""; do
  if [ -z "$v" ]; then lib=$L/libmatgcn.so; else lib=$L/libmatgcn_$v.so; fi
  export GPU_MAX_HW_QUEUES=$q MATGCN_LIB=$lib
  timeout -k 10 200 python tools/fwd_time.py --workload bm403 --train --tag "${v:-new} queues=$q" >> $out/times.log 2>&1 || exit 1
  bash tools/rehearse_rccl_1rank.sh > $out/rccl_${v:-new}_$q.log 2>&1 || exit 1
  echo "${v:-new} queues=$q  RCCL 1-rank bench: $(tail -2 $out/rccl_${v:-new}_$q.log | head -1)" >> $out/times.log
  echo "${v:-new} queues=$q  RCCL 1-rank train: $(tail -1 $out/rccl_${v:-new}_$q.log)" >> $out/times.log
done
done
grep -v amdgpu.ids $out/times.log | cut -c1-220
