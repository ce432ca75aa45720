#!/bin/bash
# round 3, lab o: k_gate16_px - the x part of layer l+1 in the gate kernel of layer l - parity, then time against the chunked x part
set -o pipefail
out=gpurun_out/r03lab_o; mkdir -p $out; rm -f $out/times.log
timeout -k 10 900 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "px_fusion or ragged or full_size or wavefront_and_serial or test_forward" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $out/pytest.log
for rep in 1 2; do
for f in 0 1; do
  for w in bm403 dc237; do
  MATGCN_PX_FUSION=$f timeout -k 10 200 python tools/fwd_time.py --workload $w --kernels --tag "px_fusion=$f rep $rep" >> $out/times.log 2>&1 || exit 1
  done
done
done
grep -v amdgpu.ids $out/times.log | sort
