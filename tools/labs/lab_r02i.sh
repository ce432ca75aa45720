#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r02i
python -m pytest tests -q -m gpu > gpurun_out/r02i/pytest.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/r02i/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r02i/smoke.log 2>&1
echo "smoke rc=$?"; tail -3 gpurun_out/r02i/smoke.log
bash tools/profile_r02.sh r02final > gpurun_out/r02i/profile.log 2>&1
echo "profile rc=$?"; cat gpurun_out/prof_r02final/pmc_kernels.txt
cp gpurun_out/prof_r02final/pmc_kernels.json profiles/r02_pmc_kernels.json
python bench.py > gpurun_out/r02i/bench.json 2> gpurun_out/r02i/bench.err
echo "bench rc=$?"
python bench.py --workload dc237 --no-train-step > gpurun_out/r02i/bench_dc237.json 2> gpurun_out/r02i/bench_dc237.err
echo "bench dc237 rc=$?"
python bench.py --workload synth4096 --no-train-step --steps 3 --warmup 1 --median 5 --median-warmup 1 > gpurun_out/r02i/bench_synth4096.json 2> gpurun_out/r02i/bench_synth4096.err
echo "bench 4096 rc=$?"
python - <<'PY'
import json
for nm in ("bench", "bench_dc237", "bench_synth4096"):
    try:
        d=json.loads(open("gpurun_out/r02i/%s.json"%nm).read().strip().splitlines()[-1])
        r=d["roofline"]
        print(nm, "ms %.3f value %.4g median %s build %s | k_mix %.1f us frac %.3f pmc: %s | bf16 %s" % (d["ms_per_step"], d["value"], d.get("median",{}).get("median_ms"), d["build_id"], r["avg_launch_ms"]*1e3, r["frac"], str(r["pmc_source"])[:60], d.get("bf16_variant",{}).get("ms_per_step")))
        print("   whole fwd executed %.1f TF frac %.3f; node kernels:" % (r["whole_forward"]["executed_tflops"], r["whole_forward"]["frac_mfma"]), {k:(round(v["avg_launch_ms"]*1e3,1), v["mfma_util_pmc"] and round(v["mfma_util_pmc"],3), v["achieved_traffic_tbs"] and round(v["achieved_traffic_tbs"],2)) for k,v in r["node_kernels"].items()})
        if "train_step" in d: print("   train", {k:round(v,2) for k,v in d["train_step"].items() if k in ("forward_ms","backward_ms","ms_per_step","backward_executed_tflops")})
        if "cpu_baseline" in d: print("   cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], "gpu/cpu", d.get("gpu_over_cpu"), "err", d["cpu_baseline"].get("gpu_vs_cpu_max_norm_err"))
    except Exception as e:
        print(nm, "FAILED", e)
PY
