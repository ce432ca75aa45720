#!/bin/bash
# round 4, lab s: the forward under an initialised RCCL process group is 18 % slower (7.97 vs 6.77 ms) while the kernels' own
# durations are unchanged.  Hypothesis: hardware queues - the runtime maps HIP streams onto GPU_MAX_HW_QUEUES (default 4)
# queues, RCCL's streams take some, the wavefront's chains then share a queue and serialise.
set -o pipefail
out=gpurun_out/r04lab_s; mkdir -p $out; rm -f $out/times.log
for q in default 8 16; do
  if [ "$q" = "default" ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  timeout -k 10 200 python tools/fwd_time.py --workload bm403 --tag "queues=$q" >> $out/times.log 2>&1 || exit 1
  timeout -k 10 200 python tools/fwd_time.py --workload bm403 --train --tag "queues=$q" >> $out/times.log 2>&1 || exit 1
  timeout -k 10 200 python tools/train_step.py bm403 6 wave 2>&1 | tail -3 | sed "s/^/queues=$q  /" >> $out/times.log || exit 1
  bash tools/rehearse_rccl_1rank.sh > $out/rccl_$q.log 2>&1 || exit 1
  echo "queues=$q  RCCL 1-rank bench: $(tail -2 $out/rccl_$q.log | head -1)" >> $out/times.log
  echo "queues=$q  RCCL 1-rank train: $(tail -1 $out/rccl_$q.log)" >> $out/times.log
done
grep -v amdgpu.ids $out/times.log | cut -c1-220
