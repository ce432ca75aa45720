#!/bin/bash
# round 4, lab b: fast gate non-linearities + x_t requested before the last K chunk (product) against the round-3 node
# kernels (r3node), and with the first mixed chunks requested after the state rows are in LDS (cas); phase stamps of both
set -o pipefail
out=gpurun_out/r04lab_b; mkdir -p $out; rm -f $out/times.log
L=multistgraph_amd/lib
./tools/labs/gates_lab > $out/gates.log 2>&1; cat $out/gates.log
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -m gpu -q -x > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
for rep in 1 2; do
for v in r3node "" cas; do
  if [ -z "$v" ]; then lib=$L/libmatgcn.so; else lib=$L/libmatgcn_$v.so; fi
  for w in bm403 dc237; do
  MATGCN_LIB=$lib timeout -k 10 200 python tools/fwd_time.py --workload $w --kernels --tag "${v:-new} rep $rep" >> $out/times.log 2>&1 || exit 1
  done
done
done
grep -v amdgpu.ids $out/times.log | sort | cut -c1-330
MATGCN_LIB=$L/libmatgcn_stamps.so timeout -k 10 200 python tools/labs/stamps_r04.py --workload bm403 > $out/stamps_new.log 2>&1 || exit 1
MATGCN_LIB=$L/libmatgcn_stamps_cas.so timeout -k 10 200 python tools/labs/stamps_r04.py --workload bm403 > $out/stamps_cas.log 2>&1 || exit 1
grep -A24 "sharing a CU" $out/stamps_new.log | cut -c1-70
