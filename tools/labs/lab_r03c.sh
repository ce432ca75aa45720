#!/bin/bash
# round 3, lab c: (node, 32-row) work items for k_gate16 / k_update16 against the 64-row build, same box
set -o pipefail
out=gpurun_out/r03lab_c; mkdir -p $out
L=multistgraph_amd/lib
for v in "" rows32_w4 rows32_w5 rows32_w6; do
  if [ -z "$v" ]; then lib=$L/libmatgcn.so; else lib=$L/libmatgcn_$v.so; fi
  MATGCN_LIB=$lib timeout -k 10 300 python -m pytest tests/test_hip_parity.py tests/test_model_gpu.py -m gpu -x -q -k "forward or atgru or encoder or predict or ragged or wavefront" > $out/pytest_${v:-rows64}.log 2>&1 || { tail -20 $out/pytest_${v:-rows64}.log; exit 1; }
  tail -1 $out/pytest_${v:-rows64}.log
  MATGCN_LIB=$lib timeout -k 10 200 python tools/fwd_time.py --kernels --tag "${v:-rows64}" >> $out/times.log 2>&1 || exit 1
  MATGCN_LIB=$lib timeout -k 10 200 python tools/fwd_time.py --workload dc237 --kernels --tag "${v:-rows64}" >> $out/times.log 2>&1 || exit 1
done
grep -v amdgpu.ids $out/times.log
