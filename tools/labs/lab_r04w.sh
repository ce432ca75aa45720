#!/bin/bash
# round 4, lab w: precision mode 2 with the hoisted x part on bf16 operands too (k_px16<NRT, true>)
set -o pipefail
out=gpurun_out/r04lab_w; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_model_gpu.py -m gpu -q -x > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $out/pytest.log
timeout -k 10 500 python bench.py --no-cpu-baseline --no-train-step --no-batch16 > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
timeout -k 10 500 python bench.py --workload dc237 --no-cpu-baseline --no-train-step --no-batch16 > $out/bench_dc237.json 2> $out/bench_dc237.err; echo "bench rc=$?"
python - <<'PY'
import json
for nm in ("bench", "bench_dc237"):
    d = json.loads(open("gpurun_out/r04lab_w/%s.json" % nm).read().strip().splitlines()[-1])
    print(nm, round(d["ms_per_step"], 3), json.dumps(d["bf16_variant"])[:900])
PY
