#!/bin/bash
# forward + backward parity subsets, the forward bench (no CPU baseline) and the step timing
set -o pipefail
mkdir -p gpurun_out/r02o
python -m pytest tests/test_hip_parity.py tests/test_model_gpu.py -x -q -m gpu > gpurun_out/r02o/pytest_fwd.log 2>&1
echo "pytest fwd rc=$?"; tail -2 gpurun_out/r02o/pytest_fwd.log
python -m pytest tests/test_backward_gpu.py -x -q -m gpu -k "bm_small or plugin or series or h0 or static or synthetic or golden or reference or gemm" > gpurun_out/r02o/pytest_bwd.log 2>&1
echo "pytest bwd rc=$?"; tail -2 gpurun_out/r02o/pytest_bwd.log
python bench.py --no-cpu-baseline --no-bf16-variant > gpurun_out/r02o/bench.json 2> gpurun_out/r02o/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r02o/bench.json").read().strip().splitlines()[-1])
print("fwd ms", d["ms_per_step"], "median", d["median"]["median_ms"], "k_mix us", d["roofline"]["avg_launch_ms"]*1e3, "train", {k:round(v,2) for k,v in d["train_step"].items() if k in ("forward_ms","backward_ms","ms_per_step")})
PY
