#!/bin/bash
# round 3, lab n: the node-adaptive weight streams of matgcn_prepare - k_prep_stream (round 1, 8 nodes per thread),
# k_prep_stream2 (pool values in registers, node after node), k_prep_mfma (the embedding contraction on the matrix cores)
set -o pipefail
out=gpurun_out/r03lab_n; mkdir -p $out; rm -f $out/times.log
L=multistgraph_amd/lib
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_model_gpu.py -m gpu -q -x > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
for rep in 1 2; do
for v in prep0 prep1 "" prep2tpw2 prep2tpw8; do
  if [ -z "$v" ]; then lib=$L/libmatgcn.so; else lib=$L/libmatgcn_$v.so; fi
  for w in bm403 dc237; do
  MATGCN_LIB=$lib timeout -k 10 200 python tools/fwd_time.py --workload $w --tag "${v:-prep2tpw4} rep $rep" >> $out/times.log 2>&1 || exit 1
  done
done
done
grep -v amdgpu.ids $out/times.log | sort
# kernel durations of the prepare kernels (serial schedule)
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp
for v in prep0 prep1 "" prep2tpw2 prep2tpw8; do
  if [ -z "$v" ]; then lib=$R/$L/libmatgcn.so; else lib=$R/$L/libmatgcn_$v.so; fi
  export MATGCN_LIB=$lib
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/stats_${v:-new} -- python3 $R/tools/fwd_time.py --workload bm403 --serial --iters 20 > $R/$out/prof_${v:-new}.log 2>&1
  S=$(find $R/$out/stats_${v:-new} -name "*kernel_stats.csv" | head -1)
  echo "== ${v:-prep2tpw4}"; grep -i "prep" "$S" | cut -d, -f1-4
  rm -rf $R/$out/stats_${v:-new}
done
