#!/bin/bash
# round 4, lab aj: arrangement search inside the library's own (highest-priority) queue pool, 4 queues, no process group:
# main 1, chain[1] 0, xpart[1] 1, aux 2 fixed; xcol x bchain over all 16 places.  usage: lab_r04aj.sh <list of tags>
set -o pipefail
out=gpurun_out/r04lab_aj; mkdir -p $out
L=$GRAFT_REPO_ROOT/multistgraph_amd/lib
for v in "$@"; do
  export MATGCN_LIB=$L/libmatgcn_$v.so
  timeout -k 10 200 python tools/fwd_time.py --workload bm403 --train --tag "$v" >> $out/times.log 2>&1 || exit 1
  timeout -k 10 200 python tools/fwd_time.py --workload bm403 --batch 16 --train --tag "$v B=16" >> $out/times.log 2>&1 || exit 1
done
grep -v amdgpu.ids $out/times.log | cut -c1-200
