#!/bin/bash
# round 3, lab k: steps per hoisted x-part chunk of the upper layers (2 / 4 / 8)
set -o pipefail
out=gpurun_out/r03lab_k; mkdir -p $out
L=multistgraph_amd/lib
for v in "" xchunk2 xchunk8; do
  if [ -z "$v" ]; then lib=$L/libmatgcn.so; else lib=$L/libmatgcn_$v.so; fi
  MATGCN_LIB=$lib timeout -k 10 300 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "test_forward or encoder" > $out/pytest_${v:-base}.log 2>&1 || { tail -20 $out/pytest_${v:-base}.log; exit 1; }
  for w in bm403 dc237; do
    MATGCN_LIB=$lib timeout -k 10 200 python tools/fwd_time.py --workload $w --tag "${v:-xchunk4}" >> $out/times.log 2>&1 || exit 1
  done
done
grep -v amdgpu.ids $out/times.log
