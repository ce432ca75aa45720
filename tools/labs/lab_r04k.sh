#!/bin/bash
# round 4, lab k: (1) pairs of step kernels on two streams (tools/labs/pairlab.hip); (2) the mix token - the graph mixes of
# both chains in one global order (matgcn_set_wavefront(2)) - against the free-running wavefront
set -o pipefail
out=gpurun_out/r04lab_k; mkdir -p $out; rm -f $out/times.log
timeout -k 10 120 ./tools/labs/pairlab > $out/pairlab.log 2>&1; cat $out/pairlab.log
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "test_forward or wavefront or serial" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $out/pytest.log
for rep in 1 2 3; do
  for w in bm403 dc237; do
  timeout -k 10 200 python tools/fwd_time.py --workload $w --tag "free-running rep $rep" >> $out/times.log 2>&1 || exit 1
  timeout -k 10 200 python tools/fwd_time.py --workload $w --token --tag "mix token rep $rep" >> $out/times.log 2>&1 || exit 1
  done
done
grep -v amdgpu.ids $out/times.log | sort | cut -c1-200
