#!/bin/bash
# round 2, lab A: parity suite on the GPU, then the co-residency experiment (node kernels at <= 96 VGPRs beside k_mix)
set -o pipefail
mkdir -p gpurun_out/r02a
python -m pytest tests -x -q -m gpu > gpurun_out/r02a/pytest.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/r02a/pytest.log
tail -5 gpurun_out/r02a/pytest.log
for tag in base new_w4 w4_r6 w5_r6; do
  lib=multistgraph_amd/lib/libmatgcn_$tag.so
  [ "$tag" = base ] && lib=multistgraph_amd/lib/libmatgcn.so
  MATGCN_LIB=$lib python bench.py --no-cpu-baseline --no-train-step --steps 40 --warmup 10 > gpurun_out/r02a/bench_${tag}_wave.json 2> gpurun_out/r02a/bench_${tag}_wave.err
  MATGCN_LIB=$lib python bench.py --no-cpu-baseline --no-train-step --steps 40 --warmup 10 --serial-streams > gpurun_out/r02a/bench_${tag}_serial.json 2> gpurun_out/r02a/bench_${tag}_serial.err
  python - "$tag" <<'PY'
import json,sys
tag=sys.argv[1]
for mode in ("wave","serial"):
    try:
        d=json.loads(open("gpurun_out/r02a/bench_%s_%s.json"%(tag,mode)).read().strip().splitlines()[-1])
        r=d.get("roofline",{})
        print(tag, mode, "ms/step %.3f"%d["ms_per_step"], "serial-kernels", r.get("serial_kernel_ms_per_forward"), "in-wavefront", r.get("kernel_ms_per_forward"))
    except Exception as e:
        print(tag, mode, "FAILED", e)
PY
done
