#!/bin/bash
# round 4, lab j: mix kernels' epilogue batched and branch-free (product) against the previous build (prevmix): parity, timing, stamps
set -o pipefail
out=gpurun_out/r04lab_j; mkdir -p $out; rm -f $out/times.log
L=multistgraph_amd/lib
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_backward_gpu.py -m gpu -q -x > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $out/pytest.log
for rep in 1 2 3; do
for v in prevmix ""; do
  if [ -z "$v" ]; then lib=$L/libmatgcn.so; else lib=$L/libmatgcn_$v.so; fi
  for w in bm403 dc237; do
  MATGCN_LIB=$lib timeout -k 10 200 python tools/fwd_time.py --workload $w --kernels --train --tag "${v:-new} rep $rep" >> $out/times.log 2>&1 || exit 1
  done
done
done
grep -v amdgpu.ids $out/times.log | sort | cut -c1-400
MATGCN_LIB=$L/libmatgcn_stamps.so timeout -k 10 200 python tools/labs/stamps_mix_r04.py > $out/stamps_mix_new.log 2>&1; grep -v amdgpu $out/stamps_mix_new.log | head -14
