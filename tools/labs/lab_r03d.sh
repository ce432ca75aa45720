#!/bin/bash
# round 3, lab d: precision mode 2 (bf16 operands for the node-wise contractions): parity + time
set -o pipefail
out=gpurun_out/r03lab_d; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "bf16 or forward or wavefront" > $out/pytest.log 2>&1 || { tail -30 $out/pytest.log; exit 1; }
tail -1 $out/pytest.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train-step --median 0 > $out/bench.json 2> $out/bench.err || { tail $out/bench.err; exit 1; }
python -c "
import json; d=json.load(open('$out/bench.json')); print('f32', d['ms_per_step']); print(json.dumps(d['bf16_variant'], indent=1))"
