#!/bin/bash
# round 4, lab o: k_px16 with its first weight groups requested before the A tile is staged (product), ring depth 6 (pxring6), against before (pxold)
set -o pipefail
out=gpurun_out/r04lab_o; mkdir -p $out; rm -f $out/times.log
L=multistgraph_amd/lib
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "test_forward or encoder" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $out/pytest.log
for rep in 1 2 3; do
for v in pxold "" pxring6; do
  if [ -z "$v" ]; then lib=$L/libmatgcn.so; else lib=$L/libmatgcn_$v.so; fi
  MATGCN_LIB=$lib timeout -k 10 200 python tools/fwd_time.py --workload bm403 --kernels --tag "${v:-new} rep $rep" >> $out/times.log 2>&1 || exit 1
done
done
grep -v amdgpu.ids $out/times.log | sort | cut -c1-330
