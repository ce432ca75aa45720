#!/bin/bash
# two ranks of bench.py on the ONE GPU of the dev box over gloo: the multi-rank code path (shards, barrier, max over
# ranks, the control-group vote, the flat-bucket gradient all-reduce) without RCCL; not a performance number
set -o pipefail
mkdir -p gpurun_out/rehearse
MATGCN_BENCH_ONE_DEVICE=1 MATGCN_BENCH_BACKEND=gloo timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
  --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 --median 10 --median-warmup 2 \
  > gpurun_out/rehearse/bench_2rank.json 2> gpurun_out/rehearse/bench_2rank.err
echo "rc=$?"; tail -c 400 gpurun_out/rehearse/bench_2rank.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/rehearse/bench_2rank.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","n_gpus","ms_per_step","scaling")}, d["config"]["global_batch"])
print(d.get("train_step"))
PY
