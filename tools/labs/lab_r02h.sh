#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r02h
python -m pytest tests/test_backward_gpu.py -x -q -m gpu > gpurun_out/r02h/pytest.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/r02h/pytest.log
python bench.py --no-cpu-baseline --no-bf16-variant --median 0 > gpurun_out/r02h/bench.json 2> gpurun_out/r02h/bench.err
echo "bench rc=$?"; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r02h/bench.json").read().strip().splitlines()[-1])
t=d["train_step"]
print("f32 fwd ms", d["ms_per_step"], "train", {k:(round(v,3) if isinstance(v,float) else v) for k,v in t.items() if k in ("forward_ms","backward_ms","optimizer_ms","ms_per_step","backward_executed_tflops")})
PY
