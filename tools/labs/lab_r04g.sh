#!/bin/bash
# round 4, lab g: the mix kernels' prologue requests in one batch (product) against the serialised prologue of round 3
# (serialprol); at N = 4096 the flushed accumulation (product) against the single chain (noflush)
set -o pipefail
out=gpurun_out/r04lab_g; mkdir -p $out; rm -f $out/times.log
L=multistgraph_amd/lib
timeout -k 10 900 python -m pytest tests/test_hip_parity.py -m gpu -q -x > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $out/pytest.log
for rep in 1 2 3; do
for v in serialprol ""; do
  if [ -z "$v" ]; then lib=$L/libmatgcn.so; else lib=$L/libmatgcn_$v.so; fi
  for w in bm403 dc237; do
  MATGCN_LIB=$lib timeout -k 10 200 python tools/fwd_time.py --workload $w --kernels --train --tag "${v:-new} rep $rep" >> $out/times.log 2>&1 || exit 1
  done
done
done
for rep in 1 2; do
for v in noflush ""; do
  if [ -z "$v" ]; then lib=$L/libmatgcn.so; else lib=$L/libmatgcn_$v.so; fi
  MATGCN_LIB=$lib timeout -k 10 300 python tools/fwd_time.py --workload synth4096 --iters 10 --kernels --tag "${v:-new} rep $rep" >> $out/times.log 2>&1 || exit 1
done
done
grep -v amdgpu.ids $out/times.log | sort | cut -c1-400
