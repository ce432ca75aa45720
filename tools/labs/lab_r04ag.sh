#!/bin/bash
# round 4, lab ag: 7 and 9 hardware queues inside an RCCL process group (8: forward 6.8 ms, training step 22.4-22.7; 10: 26)
set -o pipefail
out=gpurun_out/r04lab_ag; mkdir -p $out; rm -f $out/times.log
for q in 7 9 8; do
  export GPU_MAX_HW_QUEUES=$q
  bash tools/rehearse_rccl_1rank.sh > $out/rccl_$q.log 2>&1 || exit 1
  echo "queues=$q  RCCL 1-rank bench: $(tail -2 $out/rccl_$q.log | head -1)" >> $out/times.log
  echo "queues=$q  RCCL 1-rank train: $(tail -1 $out/rccl_$q.log)" >> $out/times.log
done
grep -v amdgpu.ids $out/times.log | cut -c1-220
