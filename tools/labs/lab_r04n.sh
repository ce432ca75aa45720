#!/bin/bash
# round 4, lab n: small batches - k_mix_c32 for B <= 16 + k_px16<NRT> (product) against the build before both (prevsb)
set -o pipefail
out=gpurun_out/r04lab_n; mkdir -p $out; rm -f $out/times.log
L=multistgraph_amd/lib
timeout -k 10 1000 python -m pytest tests/test_hip_parity.py tests/test_backward_gpu.py tests/test_model_gpu.py -m gpu -q -x > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $out/pytest.log
for rep in 1 2; do
for v in prevsb ""; do
  if [ -z "$v" ]; then lib=$L/libmatgcn.so; else lib=$L/libmatgcn_$v.so; fi
  for b in 16 32 64; do
  MATGCN_LIB=$lib timeout -k 10 200 python tools/fwd_time.py --workload bm403 --batch $b --kernels --train --tag "${v:-new} B=$b rep $rep" >> $out/times.log 2>&1 || exit 1
  done
  MATGCN_LIB=$lib timeout -k 10 200 python tools/fwd_time.py --workload dc237 --batch 16 --kernels --train --tag "${v:-new} B=16 rep $rep" >> $out/times.log 2>&1 || exit 1
done
done
grep -v amdgpu.ids $out/times.log | sort | cut -c1-400
