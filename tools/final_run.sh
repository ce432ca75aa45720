#!/bin/bash
# The round's evidence on one build, in two parts (each fits one gpurun call):
#   final_run.sh tests  [tag]   full GPU suite, smoke, bench lines of the three single-GPU workloads (+ training lines),
#                               host enqueue time, the 2-rank gloo and 1-rank RCCL rehearsals of bench.py's distributed branch
#   final_run.sh bench  [tag]   the same without the suite
#   final_run.sh profile [tag]  rocprofv3 kernel stats (serial + default schedule, training step) and the PMC passes
# Everything lands under gpurun_out/<tag>/ (and gpurun_out/prof_<tag>, pmc_train_<tag>); the summaries worth keeping are
# copied into profiles/r<round>_* by hand afterwards.
set -o pipefail
PART=${1:-tests}
TAG=${2:-final}
O=gpurun_out/$TAG
mkdir -p $O
if [ "$PART" = "tests" ] || [ "$PART" = "bench" ]; then
  if [ "$PART" = "tests" ]; then     # ("bench": the same without the suite - after a change of bench.py alone)
    python -m pytest tests -q -m gpu > $O/pytest.log 2>&1
    echo "pytest rc=$?"; tail -3 $O/pytest.log
  fi
  python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
  echo "smoke rc=$?"; tail -2 $O/smoke.log
  python bench.py > $O/bench.json 2> $O/bench.err
  echo "bench rc=$?"
  python bench.py --workload dc237 > $O/bench_dc237.json 2> $O/bench_dc237.err
  echo "bench dc237 rc=$?"
  python bench.py --workload synth4096 --steps 3 --warmup 1 --median 5 --median-warmup 1 > $O/bench_synth4096.json 2> $O/bench_synth4096.err
  echo "bench 4096 rc=$?"
  python tools/host_enqueue_time.py bm403 5 > $O/host_enqueue.log 2>&1
  tail -2 $O/host_enqueue.log
  bash tools/rehearse_2rank.sh > $O/rehearse.log 2>&1
  echo "rehearse rc=$?"; cp gpurun_out/rehearse/bench_2rank.json $O/rehearsal_2rank_gloo_one_gpu.json 2>/dev/null
  bash tools/rehearse_rccl_1rank.sh > $O/rehearse_rccl.log 2>&1
  echo "rehearse rccl rc=$?"; cp gpurun_out/rehearse/bench_rccl_1rank.json $O/rehearsal_1rank_rccl.json 2>/dev/null
else
  bash tools/profile_run.sh $TAG > $O/profile.log 2>&1
  echo "profile rc=$?"; cat gpurun_out/prof_$TAG/pmc_kernels.txt
  bash tools/pmc_train_run.sh $TAG > $O/pmc_train.log 2>&1
  echo "pmc train rc=$?"; head -24 gpurun_out/pmc_train_$TAG/train_pmc_kernels.txt
  bash tools/profile_train.sh $TAG > $O/profile_train.log 2>&1
  echo "profile train rc=$?"
  python bench.py --no-cpu-baseline > $O/bench_after_pmc.json 2> $O/bench_after_pmc.err
  echo "bench (PMC of this build replayed) rc=$?"
fi
python - $O <<'PY'
import json, sys, glob
O = sys.argv[1]
for path in sorted(glob.glob(O + "/bench*.json")):
    nm = path.split("/")[-1]
    try:
        d=json.loads(open(path).read().strip().splitlines()[-1])
        r=d["roofline"]
        print(nm, "ms %.3f value %.4g median %s build %s | k_mix %.1f us frac %.3f pmc: %s | bf16 %s / %s" % (d["ms_per_step"], d["value"], d.get("median",{}).get("median_ms"), d["build_id"], r["avg_launch_ms"]*1e3, r["frac"], str(r["pmc_source"])[:50], d.get("bf16_variant",{}).get("ms_per_step"), d.get("bf16_variant",{}).get("mix_and_node_contractions",{}).get("ms_per_step")))
        print("   whole fwd executed %.1f TF frac %.3f; node kernels:" % (r["whole_forward"]["executed_tflops"], r["whole_forward"]["frac_mfma"]), {k:(round(v["avg_launch_ms"]*1e3,1), v["mfma_util_pmc"] and round(v["mfma_util_pmc"],3), v["achieved_traffic_tbs"] and round(v["achieved_traffic_tbs"],2)) for k,v in r["node_kernels"].items()})
        if "train_step" in d: print("   train", {k:round(v,2) for k,v in d["train_step"].items() if k in ("forward_ms","backward_ms","ms_per_step","backward_executed_tflops","backward_frac_mfma","backward_over_forward_flops")})
        if "cpu_baseline" in d: print("   cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], "at 8:", d["cpu_baseline"].get("at_8_threads"), "gpu/cpu", d.get("gpu_over_cpu"), "err", d["cpu_baseline"].get("gpu_vs_cpu_max_norm_err"))
    except Exception as e:
        print(nm, "FAILED", e)
PY
