#!/bin/bash
# copies the summaries of an evidence run (tools/final_run.sh tests|profile <tag>) from gpurun_out/ into profiles/<round>_<label>_*
# usage: collect_evidence.sh <tag> <round> <label>      e.g. collect_evidence.sh r04e1 r04 e1
set -e
TAG=$1; RD=$2; LB=$3; O=gpurun_out/$TAG; P=profiles
[ -f $O/bench.json ] && cp $O/bench.json $P/${RD}_${LB}_bench.json
[ -f $O/bench_dc237.json ] && cp $O/bench_dc237.json $P/${RD}_${LB}_bench_dc237.json
[ -f $O/bench_synth4096.json ] && cp $O/bench_synth4096.json $P/${RD}_${LB}_bench_synth4096.json
[ -f $O/rehearsal_2rank_gloo_one_gpu.json ] && cp $O/rehearsal_2rank_gloo_one_gpu.json $P/${RD}_${LB}_rehearsal_2rank_gloo_one_gpu.json
[ -f $O/rehearsal_1rank_rccl.json ] && cp $O/rehearsal_1rank_rccl.json $P/${RD}_${LB}_rehearsal_1rank_rccl.json
[ -f $O/pytest.log ] && tail -4 $O/pytest.log > $P/${RD}_${LB}_pytest_gpu_tail.txt
[ -f $O/smoke.log ] && cp $O/smoke.log $P/${RD}_${LB}_smoke.log
[ -f $O/host_enqueue.log ] && cp $O/host_enqueue.log $P/${RD}_${LB}_host_enqueue.log
if [ -d gpurun_out/prof_$TAG ]; then
  cp gpurun_out/prof_$TAG/pmc_kernels.json $P/${RD}_pmc_kernels.json
  cp gpurun_out/prof_$TAG/pmc_kernels.txt $P/${RD}_${LB}_pmc_kernels.txt
  cp gpurun_out/prof_$TAG/kernel_stats_serial.csv $P/${RD}_${LB}_serial_kernel_stats.csv
  cp gpurun_out/prof_$TAG/kernel_stats_default.csv $P/${RD}_${LB}_default_kernel_stats.csv
  cp gpurun_out/pmc_train_$TAG/train_pmc_kernels.json $P/${RD}_${LB}_train_pmc_kernels.json
  cp gpurun_out/pmc_train_$TAG/train_pmc_kernels.txt $P/${RD}_${LB}_train_pmc_kernels.txt
  S=$(find gpurun_out/prof_train_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$S" ] && cp "$S" $P/${RD}_${LB}_train_step_kernel_stats.csv
  [ -f $O/bench_after_pmc.json ] && cp $O/bench_after_pmc.json $P/${RD}_${LB}_bench_with_pmc.json
fi
python - <<PY
import json, os
from multistgraph_amd import build
p="$P/${RD}_pmc_kernels.json"
if os.path.exists(p):
    d=json.load(open(p))
    print("pmc build", d.get("build_id"), "bench build", json.loads(open("$P/${RD}_${LB}_bench.json").read().strip().splitlines()[-1])["build_id"], "source now", build.source_id())
PY
ls $P | grep "${RD}_${LB}_" | wc -l
