#!/bin/bash
# round 3, lab j: k_mixr (16-byte LDS operand reads, row-major stack copy) against k_mix<1> (k-major image, 4-byte reads)
set -o pipefail
out=gpurun_out/r03lab_j; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_model_gpu.py -m gpu -x -q > $out/pytest.log 2>&1 || { tail -30 $out/pytest.log; exit 1; }
tail -1 $out/pytest.log
L=multistgraph_amd/lib
for w in bm403 dc237 synth4096; do
  it=60; [ $w = synth4096 ] && it=6
  MATGCN_LIB=$L/libmatgcn_kmajor.so timeout -k 10 300 python tools/fwd_time.py --workload $w --iters $it --kernels --tag "k_mix<1>" >> $out/times.log 2>&1 || exit 1
  timeout -k 10 300 python tools/fwd_time.py --workload $w --iters $it --kernels --tag "k_mixr" >> $out/times.log 2>&1 || exit 1
done
grep -v amdgpu.ids $out/times.log
