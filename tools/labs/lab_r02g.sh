#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r02g
python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "bf16 or test_forward" > gpurun_out/r02g/pytest.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/r02g/pytest.log
python bench.py --no-cpu-baseline --no-train-step > gpurun_out/r02g/bench.json 2> gpurun_out/r02g/bench.err
echo "bench rc=$?"; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r02g/bench.json").read().strip().splitlines()[-1])
print("f32 ms", d["ms_per_step"], "bf16", d.get("bf16_variant"))
PY
