#!/bin/bash
# training step at the other two single-GPU shapes of BASELINE.json (DC 237 nodes, synthetic 4096 nodes)
set -o pipefail
mkdir -p gpurun_out/r02p
for wl in dc237 synth4096; do
  python bench.py --workload $wl --no-cpu-baseline --no-bf16-variant --median 0 --steps 3 --warmup 1 > gpurun_out/r02p/bench_$wl.json 2> gpurun_out/r02p/bench_$wl.err
  echo "$wl rc=$?"
  python - $wl <<'PY'
import json, sys
d=json.loads(open("gpurun_out/r02p/bench_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
t=d.get("train_step", {})
print(sys.argv[1], "fwd ms %.2f" % d["ms_per_step"], {k:(round(v,2) if isinstance(v,float) else v) for k,v in t.items() if k in ("forward_ms","backward_ms","ms_per_step","loss_first","loss_last","backward_executed_tflops")})
PY
done
