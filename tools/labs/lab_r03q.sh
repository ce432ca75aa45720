#!/bin/bash
# round 3, lab q: the tail of the backward - layer 0's narrow transposed mix on k_mix, narrow residual-cell x columns
set -o pipefail
out=gpurun_out/r03lab_q; mkdir -p $out; rm -f $out/times.log
L=multistgraph_amd/lib
timeout -k 10 1000 python -m pytest tests/test_backward_gpu.py -m gpu -q -x > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
for rep in 1 2; do
for v in base ""; do
  if [ -z "$v" ]; then lib=$L/libmatgcn.so; else lib=$L/libmatgcn_$v.so; fi
  for w in bm403 dc237; do
  MATGCN_LIB=$lib timeout -k 10 300 python tools/fwd_time.py --workload $w --train --tag "${v:-new} rep $rep" >> $out/times.log 2>&1 || exit 1
  done
done
done
grep -v amdgpu.ids $out/times.log | sort
