#!/bin/bash
# round 4, lab r: k_mix reads the next K-tile's fragments before this tile's MFMAs (new) against the round-3 loop (mixr3)
set -o pipefail
out=gpurun_out/r04lab_r; mkdir -p $out; rm -f $out/times.log
L=multistgraph_amd/lib
timeout -k 10 1000 python -m pytest tests/test_hip_parity.py tests/test_model_gpu.py -m gpu -q -x > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $out/pytest.log
for rep in 1 2 3; do
for v in mixr3 ""; do
  if [ -z "$v" ]; then lib=$L/libmatgcn.so; else lib=$L/libmatgcn_$v.so; fi
  MATGCN_LIB=$lib timeout -k 10 200 python tools/fwd_time.py --workload bm403 --tag "${v:-new} rep $rep" >> $out/times.log 2>&1 || exit 1
  MATGCN_LIB=$lib timeout -k 10 200 python tools/fwd_time.py --workload dc237 --tag "${v:-new} rep $rep" >> $out/times.log 2>&1 || exit 1
done
done
for v in mixr3 ""; do
  if [ -z "$v" ]; then lib=$L/libmatgcn.so; else lib=$L/libmatgcn_$v.so; fi
  MATGCN_LIB=$lib timeout -k 10 300 python tools/fwd_time.py --workload synth4096 --iters 5 --tag "${v:-new}" >> $out/times.log 2>&1 || exit 1
done
grep -v amdgpu.ids $out/times.log | sort | cut -c1-200
cd /tmp && export TMPDIR=/tmp
for v in mixr3 ""; do
  if [ -z "$v" ]; then lib=$L/libmatgcn.so; else lib=$L/libmatgcn_$v.so; fi
  export MATGCN_LIB=$GRAFT_REPO_ROOT/$lib
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof_${v:-new} -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train-step --no-batch16 --serial-streams > $GRAFT_REPO_ROOT/$out/prof_${v:-new}.log 2>&1 || exit 1
  echo "== ${v:-new}"; python3 - <<PY
import csv,glob
f=glob.glob("$GRAFT_REPO_ROOT/$out/prof_${v:-new}/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if float(r["Percentage"])>1: print("%-60s %6s calls avg %8.2f us  %5.1f %%"%(r["Name"][:60],r["Calls"],float(r["AverageNs"])/1e3,float(r["Percentage"])))
PY
done
