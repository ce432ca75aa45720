#!/bin/bash
# round 4, lab ad: k_px16 with the steps of a chunk in ONE tile per node when they fit (B = 16 / 32; new) against one work item
# per (node, step) (pxnomerge)
set -o pipefail
out=gpurun_out/r04lab_ad; mkdir -p $out; rm -f $out/times.log
L=multistgraph_amd/lib
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_model_gpu.py tests/test_backward_gpu.py -m gpu -q -x -k "ragged or synthetic_shapes or bf16 or golden or batch" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $out/pytest.log
for rep in 1 2 3; do
for v in pxnomerge ""; do
  if [ -z "$v" ]; then lib=$L/libmatgcn.so; else lib=$L/libmatgcn_$v.so; fi
  MATGCN_LIB=$lib timeout -k 10 200 python tools/fwd_time.py --workload bm403 --batch 16 --train --tag "${v:-new} B=16 rep $rep" >> $out/times.log 2>&1 || exit 1
  MATGCN_LIB=$lib timeout -k 10 200 python tools/fwd_time.py --workload dc237 --batch 16 --train --tag "${v:-new} B=16 rep $rep" >> $out/times.log 2>&1 || exit 1
  MATGCN_LIB=$lib timeout -k 10 200 python tools/fwd_time.py --workload bm403 --batch 32 --train --tag "${v:-new} B=32 rep $rep" >> $out/times.log 2>&1 || exit 1
done
done
MATGCN_LIB=$L/libmatgcn_pxnomerge.so timeout -k 10 200 python tools/fwd_time.py --workload bm403 --batch 16 --kernels --tag "pxnomerge B=16" >> $out/times.log 2>&1
timeout -k 10 200 python tools/fwd_time.py --workload bm403 --batch 16 --kernels --tag "new B=16" >> $out/times.log 2>&1
MATGCN_LIB=$L/libmatgcn_pxnomerge.so timeout -k 10 200 python tools/fwd_time.py --workload bm403 --batch 32 --kernels --tag "pxnomerge B=32" >> $out/times.log 2>&1
timeout -k 10 200 python tools/fwd_time.py --workload bm403 --batch 32 --kernels --tag "new B=32" >> $out/times.log 2>&1
grep -v amdgpu.ids $out/times.log | sort | cut -c1-250
