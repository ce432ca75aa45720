#!/bin/bash
# round 3, lab v: k_head with coalesced slab loads through per-wave LDS, k_build_xa0 per node through LDS - parity, then time
set -o pipefail
out=gpurun_out/r03lab_v; mkdir -p $out; rm -f $out/times.log
L=multistgraph_amd/lib
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_model_gpu.py -m gpu -q -x > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
for rep in 1 2 3; do
for v in base ""; do
  if [ -z "$v" ]; then lib=$L/libmatgcn.so; else lib=$L/libmatgcn_$v.so; fi
  for w in bm403 dc237; do
  MATGCN_LIB=$lib timeout -k 10 200 python tools/fwd_time.py --workload $w --kernels --tag "${v:-new} rep $rep" >> $out/times.log 2>&1 || exit 1
  done
done
done
grep -v amdgpu.ids $out/times.log | sort | cut -c1-260
