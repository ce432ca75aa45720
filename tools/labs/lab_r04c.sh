#!/bin/bash
# round 4, lab c: node-kernel epilogues restructured (uniform wave index, batched LDS reads, z*h in the row sweep, saved
# activations as float4 rows) against the round-3 node kernels (r3node = precise gates, x_t late off, but THIS epilogue
# structure is in both - the round-3 numbers are lab b's); cas = first mixed chunks after the state rows; dev kernarg
set -o pipefail
out=gpurun_out/r04lab_c; mkdir -p $out; rm -f $out/times.log
L=multistgraph_amd/lib
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_backward_gpu.py -m gpu -q -x > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
for rep in 1 2; do
for v in "" cas; do
  if [ -z "$v" ]; then lib=$L/libmatgcn.so; else lib=$L/libmatgcn_$v.so; fi
  for w in bm403 dc237; do
  MATGCN_LIB=$lib timeout -k 10 200 python tools/fwd_time.py --workload $w --kernels --train --tag "${v:-new} rep $rep" >> $out/times.log 2>&1 || exit 1
  done
done
HIP_FORCE_DEV_KERNARG=1 timeout -k 10 200 python tools/fwd_time.py --workload bm403 --kernels --train --tag "new devkernarg rep $rep" >> $out/times.log 2>&1 || exit 1
done
grep -v amdgpu.ids $out/times.log | sort | cut -c1-400
MATGCN_LIB=$L/libmatgcn_stamps.so timeout -k 10 200 python tools/labs/stamps_r04.py --workload bm403 > $out/stamps_new.log 2>&1 || exit 1
grep -A24 "sharing a CU" $out/stamps_new.log | cut -c1-70
