#!/bin/bash
# round 4, lab ac: k_mix_c32 (B <= 16) with two K-tiles per barrier (new) against one (c32kt1)
set -o pipefail
out=gpurun_out/r04lab_ac; mkdir -p $out; rm -f $out/times.log
L=multistgraph_amd/lib
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_model_gpu.py -m gpu -q -x > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $out/pytest.log
for rep in 1 2 3; do
for v in c32kt1 ""; do
  if [ -z "$v" ]; then lib=$L/libmatgcn.so; else lib=$L/libmatgcn_$v.so; fi
  MATGCN_LIB=$lib timeout -k 10 200 python tools/fwd_time.py --workload bm403 --batch 16 --train --tag "${v:-new} B=16 rep $rep" >> $out/times.log 2>&1 || exit 1
  MATGCN_LIB=$lib timeout -k 10 200 python tools/fwd_time.py --workload dc237 --batch 16 --train --tag "${v:-new} B=16 rep $rep" >> $out/times.log 2>&1 || exit 1
  MATGCN_LIB=$lib timeout -k 10 200 python tools/fwd_time.py --workload bm403 --batch 8 --tag "${v:-new} B=8 rep $rep" >> $out/times.log 2>&1 || exit 1
done
done
MATGCN_LIB=$L/libmatgcn_c32kt1.so timeout -k 10 200 python tools/fwd_time.py --workload bm403 --batch 16 --kernels --tag "c32kt1 B=16" >> $out/times.log 2>&1
timeout -k 10 200 python tools/fwd_time.py --workload bm403 --batch 16 --kernels --tag "new B=16" >> $out/times.log 2>&1
grep -v amdgpu.ids $out/times.log | sort | cut -c1-250
