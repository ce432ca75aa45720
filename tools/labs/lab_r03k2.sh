#!/bin/bash
# round 3, lab k2: steps per hoisted x-part chunk 1 / 2 / 3 / 4, interleaved twice to see the run-to-run spread
set -o pipefail
out=gpurun_out/r03lab_k2; mkdir -p $out
L=multistgraph_amd/lib
for rep in 1 2; do
for v in "" xchunk1 xchunk2 xchunk3; do
  if [ -z "$v" ]; then lib=$L/libmatgcn.so; else lib=$L/libmatgcn_$v.so; fi
  for w in bm403 dc237; do
    MATGCN_LIB=$lib timeout -k 10 200 python tools/fwd_time.py --workload $w --tag "${v:-xchunk4} rep $rep" >> $out/times.log 2>&1 || exit 1
  done
done
done
MATGCN_LIB=$L/libmatgcn_xchunk1.so timeout -k 10 300 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "test_forward or encoder or wavefront" > $out/pytest_xchunk1.log 2>&1 || { tail -20 $out/pytest_xchunk1.log; exit 1; }
grep -v amdgpu.ids $out/times.log | sort
