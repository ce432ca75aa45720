#!/bin/bash
# round 4, lab p: 32-row work items for the TRAINING forward's node kernels too (B <= 32 or N <= 256) (new) against before (prevsv)
set -o pipefail
out=gpurun_out/r04lab_p; mkdir -p $out; rm -f $out/times.log
L=multistgraph_amd/lib
timeout -k 10 1000 python -m pytest tests/test_backward_gpu.py tests/test_model_gpu.py tests/test_windows.py -m gpu -q -x > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $out/pytest.log
for rep in 1 2; do
for v in prevsv ""; do
  if [ -z "$v" ]; then lib=$L/libmatgcn.so; else lib=$L/libmatgcn_$v.so; fi
  MATGCN_LIB=$lib timeout -k 10 200 python tools/fwd_time.py --workload bm403 --batch 16 --train --tag "${v:-new} B=16 rep $rep" >> $out/times.log 2>&1 || exit 1
  MATGCN_LIB=$lib timeout -k 10 200 python tools/fwd_time.py --workload dc237 --batch 16 --train --tag "${v:-new} B=16 rep $rep" >> $out/times.log 2>&1 || exit 1
  MATGCN_LIB=$lib timeout -k 10 200 python tools/fwd_time.py --workload dc237 --train --tag "${v:-new} B=64 rep $rep" >> $out/times.log 2>&1 || exit 1
done
done
grep -v amdgpu.ids $out/times.log | sort | cut -c1-200
