#!/bin/bash
# copies the summaries of an evidence run (tools/final_r03.sh tests|profile <tag>) from gpurun_out/ into profiles/r03_<label>_*
# usage: collect_evidence.sh <tag> <label>      e.g. collect_evidence.sh r03e5 e5
set -e
TAG=$1; LB=$2; O=gpurun_out/$TAG; P=profiles
cp $O/bench.json $P/r03_${LB}_bench.json
cp $O/bench_dc237.json $P/r03_${LB}_bench_dc237.json
cp $O/bench_synth4096.json $P/r03_${LB}_bench_synth4096.json
cp $O/rehearsal_2rank_gloo_one_gpu.json $P/r03_${LB}_rehearsal_2rank_gloo_one_gpu.json
tail -4 $O/pytest.log > $P/r03_${LB}_pytest_gpu_tail.txt
cp $O/smoke.log $P/r03_${LB}_smoke.log
cp $O/host_enqueue.log $P/r03_${LB}_host_enqueue.log
if [ -d gpurun_out/prof_$TAG ]; then
  cp gpurun_out/prof_$TAG/pmc_kernels.json $P/r03_pmc_kernels.json
  cp gpurun_out/prof_$TAG/pmc_kernels.txt $P/r03_${LB}_pmc_kernels.txt
  cp gpurun_out/prof_$TAG/kernel_stats_serial.csv $P/r03_${LB}_serial_kernel_stats.csv
  cp gpurun_out/prof_$TAG/kernel_stats_default.csv $P/r03_${LB}_default_kernel_stats.csv
  cp gpurun_out/pmc_train_$TAG/train_pmc_kernels.json $P/r03_${LB}_train_pmc_kernels.json
  cp gpurun_out/pmc_train_$TAG/train_pmc_kernels.txt $P/r03_${LB}_train_pmc_kernels.txt
  S=$(find gpurun_out/prof_train_$TAG -name "*kernel_stats.csv" | head -1); cp "$S" $P/r03_${LB}_train_step_kernel_stats.csv
  cp $O/bench_after_pmc.json $P/r03_${LB}_bench_with_pmc.json
fi
python - <<PY
import json
from multistgraph_amd import build
d=json.load(open("$P/r03_pmc_kernels.json"))
print("pmc build", d.get("build_id"), "bench build", json.loads(open("$P/r03_${LB}_bench.json").read().strip().splitlines()[-1])["build_id"], "source now", build.source_id())
PY
ls $P | grep "r03_${LB}_" | wc -l
