#!/bin/bash
# round 4, lab y: the activations forward_train saves (z, r, hc, z2, r2, hc2, the dropout-masked sequence) written through (sc1; new)
# against plain stores that leave dirty L2 lines for the end-of-kernel flush (saveplain)
set -o pipefail
out=gpurun_out/r04lab_y; mkdir -p $out; rm -f $out/times.log
L=multistgraph_amd/lib
timeout -k 10 900 python -m pytest tests/test_backward_gpu.py -m gpu -q -x > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $out/pytest.log
for rep in 1 2 3; do
for v in saveplain ""; do
  if [ -z "$v" ]; then lib=$L/libmatgcn.so; else lib=$L/libmatgcn_$v.so; fi
  MATGCN_LIB=$lib timeout -k 10 200 python tools/fwd_time.py --workload bm403 --train --tag "${v:-new} rep $rep" >> $out/times.log 2>&1 || exit 1
  MATGCN_LIB=$lib timeout -k 10 200 python tools/fwd_time.py --workload dc237 --train --tag "${v:-new} rep $rep" >> $out/times.log 2>&1 || exit 1
  MATGCN_LIB=$lib timeout -k 10 200 python tools/fwd_time.py --workload bm403 --batch 16 --train --tag "${v:-new} B=16 rep $rep" >> $out/times.log 2>&1 || exit 1
done
done
grep -v amdgpu.ids $out/times.log | sort | cut -c1-220
