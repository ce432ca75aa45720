#!/bin/bash
# round 4, lab ah: the hot entry points on a library stream of their own (`main`) instead of the caller's, created first (new), behind
# layer 1's pair (main1) or last (main2), against the caller's stream (nomain) - without a process group at 4 queues, and inside an RCCL
# process group at 4 and 8 queues
set -o pipefail
out=gpurun_out/r04lab_ah; mkdir -p $out; rm -f $out/times.log
L=$GRAFT_REPO_ROOT/multistgraph_amd/lib
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_model_gpu.py tests/test_windows.py -m gpu -q -x > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $out/pytest.log
for v in nomain "" main1 main2; do
  if [ -z "$v" ]; then lib=$L/libmatgcn.so; else lib=$L/libmatgcn_$v.so; fi
  export MATGCN_LIB=$lib
  unset GPU_MAX_HW_QUEUES
  timeout -k 10 200 python tools/fwd_time.py --workload bm403 --train --tag "${v:-main0} no-PG q=4" >> $out/times.log 2>&1 || exit 1
  timeout -k 10 200 python tools/fwd_time.py --workload bm403 --batch 16 --train --tag "${v:-main0} no-PG q=4 B=16" >> $out/times.log 2>&1 || exit 1
  for q in 4 8; do
    export GPU_MAX_HW_QUEUES=$q
    bash tools/rehearse_rccl_1rank.sh > $out/rccl_${v:-main0}_$q.log 2>&1 || exit 1
    echo "${v:-main0} RCCL q=$q bench: $(tail -2 $out/rccl_${v:-main0}_$q.log | head -1)" >> $out/times.log
    echo "${v:-main0} RCCL q=$q train: $(tail -1 $out/rccl_${v:-main0}_$q.log)" >> $out/times.log
  done
done
grep -v amdgpu.ids $out/times.log | cut -c1-230
