#!/bin/bash
# round 4, lab e: + k_chain_gate_node (batched requests in the gate block of the chain) - gradient parity (incl. the new
# embed / head-layout cases), training-step timing against r3chain, serial kernel stats
set -o pipefail
out=gpurun_out/r04lab_e; mkdir -p $out; rm -f $out/times.log
L=multistgraph_amd/lib
timeout -k 10 1000 python -m pytest tests/test_backward_gpu.py -m gpu -q -x > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
for rep in 1 2; do
for v in r3chain ""; do
  if [ -z "$v" ]; then lib=$L/libmatgcn.so; else lib=$L/libmatgcn_$v.so; fi
  for w in bm403 dc237; do
  MATGCN_LIB=$lib timeout -k 10 200 python tools/fwd_time.py --workload $w --train --tag "${v:-new} rep $rep" >> $out/times.log 2>&1 || exit 1
  done
done
done
grep -v amdgpu.ids $out/times.log | sort | cut -c1-200
bash tools/labs/prof_train_serial.sh new_e | head -16
