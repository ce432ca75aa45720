#!/bin/bash
# rocprofv3 kernel trace + HIP runtime API trace of three training steps: shows WHEN the host enqueued each launch
# next to when the GPU ran it (host-bound gaps in the two-stream backward).  usage: profile_train_hip.sh <tag>
set -o pipefail
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_trainhip_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d $OUT -- python3 $R/tools/train_step.py bm403 3 > $OUT/steps.log 2> $OUT/err.log
echo "rc=$?"; ls $OUT/*/ | head; tail -4 $OUT/steps.log
