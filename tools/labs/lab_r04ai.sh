#!/bin/bash
# round 4, lab ai: every library stream at the device's highest priority (a hardware-queue pool of the library's own) and the hot entry
# points on the library's `main` stream, created last behind 0-3 unused streams (= at each of the four places of the round robin);
# noprio = the same without the priority.  Without a process group and inside an RCCL process group, 4 queues (and 8).
set -o pipefail
out=gpurun_out/r04lab_ai; mkdir -p $out; rm -f $out/times.log
L=$GRAFT_REPO_ROOT/multistgraph_amd/lib
for v in "" pad1 pad2 pad3 noprio; do
  if [ -z "$v" ]; then lib=$L/libmatgcn.so; else lib=$L/libmatgcn_$v.so; fi
  export MATGCN_LIB=$lib
  unset GPU_MAX_HW_QUEUES
  timeout -k 10 200 python tools/fwd_time.py --workload bm403 --train --tag "${v:-pad0} no-PG q=4" >> $out/times.log 2>&1 || exit 1
  timeout -k 10 200 python tools/fwd_time.py --workload bm403 --batch 16 --train --tag "${v:-pad0} no-PG q=4 B=16" >> $out/times.log 2>&1 || exit 1
  for q in 4 8; do
    export GPU_MAX_HW_QUEUES=$q
    bash tools/rehearse_rccl_1rank.sh > $out/rccl_${v:-pad0}_$q.log 2>&1 || exit 1
    echo "${v:-pad0} RCCL q=$q bench: $(tail -2 $out/rccl_${v:-pad0}_$q.log | head -1)" >> $out/times.log
    echo "${v:-pad0} RCCL q=$q train: $(tail -1 $out/rccl_${v:-pad0}_$q.log)" >> $out/times.log
  done
done
grep -v amdgpu.ids $out/times.log | cut -c1-230
