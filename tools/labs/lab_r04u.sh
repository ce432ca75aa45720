#!/bin/bash
# round 4, lab u: under RCCL, the library's streams created BEFORE the process group (matgcn_init_streams; new default of
# bench.py) against after it (MATGCN_BENCH_STREAMS_AFTER=1), GPU_MAX_HW_QUEUES = 8 (bench.py's default) and 4
set -o pipefail
out=gpurun_out/r04lab_u; mkdir -p $out; rm -f $out/times.log
timeout -k 10 300 python -m pytest tests/test_sharding.py tests/test_host_logic.py -q -m gpu -x > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $out/pytest.log
for q in 8 4; do
for after in 0 1; do
  export GPU_MAX_HW_QUEUES=$q MATGCN_BENCH_STREAMS_AFTER=$after
  bash tools/rehearse_rccl_1rank.sh > $out/rccl_${q}_$after.log 2>&1 || exit 1
  echo "queues=$q streams_after=$after  RCCL 1-rank bench: $(tail -2 $out/rccl_${q}_$after.log | head -1)" >> $out/times.log
  echo "queues=$q streams_after=$after  RCCL 1-rank train: $(tail -1 $out/rccl_${q}_$after.log)" >> $out/times.log
done
done
unset GPU_MAX_HW_QUEUES MATGCN_BENCH_STREAMS_AFTER
timeout -k 10 200 python tools/fwd_time.py --workload bm403 --train --tag "no process group" >> $out/times.log 2>&1
grep -v amdgpu.ids $out/times.log | cut -c1-230
