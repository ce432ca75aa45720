#!/bin/bash
# everything the round's evidence consists of, on one build: full GPU suite, smoke, profiler passes, bench lines
set -o pipefail
TAG=${1:-r03final}
O=gpurun_out/$TAG
mkdir -p $O
python -m pytest tests -q -m gpu > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -3 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
echo "smoke rc=$?"; tail -2 $O/smoke.log
bash tools/profile_r03.sh $TAG > $O/profile.log 2>&1
echo "profile rc=$?"; cat gpurun_out/prof_$TAG/pmc_kernels.txt
cp gpurun_out/prof_$TAG/pmc_kernels.json profiles/r03_pmc_kernels.json
bash tools/pmc_train_r03.sh $TAG > $O/pmc_train.log 2>&1
echo "pmc train rc=$?"; head -24 gpurun_out/pmc_train_$TAG/train_pmc_kernels.txt
bash tools/profile_train.sh $TAG > $O/profile_train.log 2>&1
echo "profile train rc=$?"
python bench.py > $O/bench.json 2> $O/bench.err
echo "bench rc=$?"
python bench.py --workload dc237 --no-train-step > $O/bench_dc237.json 2> $O/bench_dc237.err
echo "bench dc237 rc=$?"
python bench.py --workload synth4096 --no-train-step --steps 3 --warmup 1 --median 5 --median-warmup 1 > $O/bench_synth4096.json 2> $O/bench_synth4096.err
echo "bench 4096 rc=$?"
python tools/host_enqueue_time.py bm403 5 > $O/host_enqueue.log 2>&1
tail -2 $O/host_enqueue.log
python - $O <<'PY'
import json, sys
O = sys.argv[1]
for nm in ("bench", "bench_dc237", "bench_synth4096"):
    try:
        d=json.loads(open("%s/%s.json"%(O,nm)).read().strip().splitlines()[-1])
        r=d["roofline"]
        print(nm, "ms %.3f value %.4g median %s build %s | k_mix %.1f us frac %.3f pmc: %s | bf16 %s" % (d["ms_per_step"], d["value"], d.get("median",{}).get("median_ms"), d["build_id"], r["avg_launch_ms"]*1e3, r["frac"], str(r["pmc_source"])[:60], d.get("bf16_variant",{}).get("ms_per_step")))
        print("   whole fwd executed %.1f TF frac %.3f; node kernels:" % (r["whole_forward"]["executed_tflops"], r["whole_forward"]["frac_mfma"]), {k:(round(v["avg_launch_ms"]*1e3,1), v["mfma_util_pmc"] and round(v["mfma_util_pmc"],3), v["achieved_traffic_tbs"] and round(v["achieved_traffic_tbs"],2)) for k,v in r["node_kernels"].items()})
        if "train_step" in d: print("   train", {k:round(v,2) for k,v in d["train_step"].items() if k in ("forward_ms","backward_ms","ms_per_step","backward_executed_tflops")})
        if "cpu_baseline" in d: print("   cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], "gpu/cpu", d.get("gpu_over_cpu"), "err", d["cpu_baseline"].get("gpu_vs_cpu_max_norm_err"))
    except Exception as e:
        print(nm, "FAILED", e)
PY
