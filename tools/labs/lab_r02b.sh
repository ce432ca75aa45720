#!/bin/bash
# round 2, lab B: pipelined node kernels with late PX - parity subset, then bench (default = 4 waves/SIMD, variant 5)
set -o pipefail
mkdir -p gpurun_out/r02b
python -m pytest tests/test_hip_parity.py tests/test_model_gpu.py tests/test_windows.py -x -q -m gpu -k "not synth4096" > gpurun_out/r02b/pytest.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/r02b/pytest.log
python -m pytest tests/test_backward_gpu.py -x -q -m gpu -k "reference_autograd or plugin_training or synthetic_shapes" > gpurun_out/r02b/pytest_bwd.log 2>&1
echo "pytest bwd rc=$?"; tail -3 gpurun_out/r02b/pytest_bwd.log
for tag in base new_w5; do
  lib=multistgraph_amd/lib/libmatgcn_$tag.so
  [ "$tag" = base ] && lib=multistgraph_amd/lib/libmatgcn.so
  for mode in wave serial; do
    extra=""; [ $mode = serial ] && extra="--serial-streams"
    MATGCN_LIB=$lib python bench.py --no-cpu-baseline --no-train-step --steps 40 --warmup 10 $extra > gpurun_out/r02b/bench_${tag}_$mode.json 2> gpurun_out/r02b/bench_${tag}_$mode.err
  done
  python - "$tag" <<'PY'
import json,sys
tag=sys.argv[1]
for mode in ("wave","serial"):
    try:
        d=json.loads(open("gpurun_out/r02b/bench_%s_%s.json"%(tag,mode)).read().strip().splitlines()[-1])
        r=d.get("roofline",{})
        print(tag, mode, "ms/step %.3f"%d["ms_per_step"], "serial-kernels", r.get("serial_kernel_ms_per_forward"))
    except Exception as e:
        print(tag, mode, "FAILED", e)
PY
done
