#!/bin/bash
# round 4, lab a: in-kernel phase stamps of k_gate16 / k_update16 (lib/libmatgcn_stamps.so, -DNODE_LAB_STAMPS) inside one
# serial forward at BM 403 and DC 237, next to the product's per-kernel launch averages on the same box
set -o pipefail
out=gpurun_out/r04lab_a; mkdir -p $out
L=multistgraph_amd/lib
timeout -k 10 200 python tools/fwd_time.py --workload bm403 --kernels --tag "product" > $out/times.log 2>&1 || exit 1
MATGCN_LIB=$L/libmatgcn_stamps.so timeout -k 10 200 python tools/labs/stamps_r04.py --workload bm403 > $out/stamps_bm403.log 2>&1 || { tail -20 $out/stamps_bm403.log; exit 1; }
MATGCN_LIB=$L/libmatgcn_stamps.so timeout -k 10 200 python tools/labs/stamps_r04.py --workload bm403 --launch 30 > $out/stamps_bm403_l1.log 2>&1 || exit 1
MATGCN_LIB=$L/libmatgcn_stamps.so timeout -k 10 200 python tools/labs/stamps_r04.py --workload dc237 > $out/stamps_dc237.log 2>&1 || exit 1
grep -v amdgpu.ids $out/times.log | cut -c1-400
grep -v amdgpu.ids $out/stamps_bm403.log
