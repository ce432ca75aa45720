#!/bin/bash
# round 4, lab ak: the shipped own-pool mode (matgcn_set_stream_pool(1): bench.py inside a process group) against the default streams
# inside the process group (MATGCN_BENCH_SHARED_POOL=1) at 4 and 8 queues; the single-process default for reference
set -o pipefail
out=gpurun_out/r04lab_ak; mkdir -p $out; rm -f $out/times.log
timeout -k 10 600 python -m pytest tests/test_sharding.py tests/test_host_logic.py tests/test_model_gpu.py -m gpu -q -x > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $out/pytest.log
for q in 4 8; do
for shared in 0 1; do
  export GPU_MAX_HW_QUEUES=$q MATGCN_BENCH_SHARED_POOL=$shared
  bash tools/rehearse_rccl_1rank.sh > $out/rccl_${q}_$shared.log 2>&1 || exit 1
  echo "queues=$q shared_pool=$shared  RCCL 1-rank bench: $(tail -2 $out/rccl_${q}_$shared.log | head -1)" >> $out/times.log
  echo "queues=$q shared_pool=$shared  RCCL 1-rank train: $(tail -1 $out/rccl_${q}_$shared.log)" >> $out/times.log
done
done
unset GPU_MAX_HW_QUEUES MATGCN_BENCH_SHARED_POOL
timeout -k 10 200 python tools/fwd_time.py --workload bm403 --train --tag "single process, default streams" >> $out/times.log 2>&1
timeout -k 10 200 python tools/train_loop_wall.py bm403 64 >> $out/times.log 2>&1
timeout -k 10 200 python tools/train_loop_wall.py bm403 16 >> $out/times.log 2>&1
grep -v amdgpu.ids $out/times.log | cut -c1-230
