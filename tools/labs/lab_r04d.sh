#!/bin/bash
# round 4, lab d: k_chain_res_node (row-local chain part + the update block's node contraction, one workgroup per node)
# against round 3's pair k_chain_res_fused + k_chain_node<false, 64> (r3chain): gradient parity, then training-step timing
set -o pipefail
out=gpurun_out/r04lab_d; mkdir -p $out; rm -f $out/times.log
L=multistgraph_amd/lib
timeout -k 10 900 python -m pytest tests/test_backward_gpu.py -m gpu -q -x > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
for rep in 1 2; do
for v in r3chain ""; do
  if [ -z "$v" ]; then lib=$L/libmatgcn.so; else lib=$L/libmatgcn_$v.so; fi
  for w in bm403 dc237; do
  MATGCN_LIB=$lib timeout -k 10 200 python tools/fwd_time.py --workload $w --train --tag "${v:-new} rep $rep" >> $out/times.log 2>&1 || exit 1
  MATGCN_LIB=$lib timeout -k 10 200 python tools/fwd_time.py --workload $w --train --serial --tag "${v:-new} serial rep $rep" >> $out/times.log 2>&1 || exit 1
  done
done
done
grep -v amdgpu.ids $out/times.log | sort | cut -c1-200
