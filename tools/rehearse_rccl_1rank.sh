#!/bin/bash
# bench.py's multi-rank branch with the ONE rank a one-GPU box has, over the real backend (nccl = RCCL): process group on
# the device, barriers, the MAX / SUM reductions of the timing, the flat-bucket gradient all-reduce of the training step -
# RCCL calls on GPU tensors instead of the gloo rehearsal's, the library streams in their own queue pool; not a scaling number (n_gpus = 1)
set -o pipefail
mkdir -p gpurun_out/rehearse
MATGCN_BENCH_FORCE_DIST=1 timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 \
  --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 5 --warmup 2 --median 10 --median-warmup 2 --no-cpu-baseline --no-batch16 \
  > gpurun_out/rehearse/bench_rccl_1rank.json 2> gpurun_out/rehearse/bench_rccl_1rank.err
echo "rc=$?"; tail -c 400 gpurun_out/rehearse/bench_rccl_1rank.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/rehearse/bench_rccl_1rank.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","n_gpus","ms_per_step","scaling")}, d["config"].get("global_batch"))
print({k: d["train_step"].get(k) for k in ("forward_ms","backward_ms","grad_allreduce_ms","ms_per_step")})
PY
