#!/bin/bash
# round 4, lab l: (node, 32-row) work items picked at RUN time for graphs of at most 256 nodes (product) against 64-row items (rows64), DC 237
set -o pipefail
out=gpurun_out/r04lab_l; mkdir -p $out; rm -f $out/times.log
L=multistgraph_amd/lib
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "dc237 or tiny_multi_uni_c2 or serial or wavefront" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $out/pytest.log
for rep in 1 2 3; do
for v in rows64 ""; do
  if [ -z "$v" ]; then lib=$L/libmatgcn.so; else lib=$L/libmatgcn_$v.so; fi
  MATGCN_LIB=$lib timeout -k 10 200 python tools/fwd_time.py --workload dc237 --kernels --tag "${v:-rows32 at run time} rep $rep" >> $out/times.log 2>&1 || exit 1
done
done
grep -v amdgpu.ids $out/times.log | sort | cut -c1-330
