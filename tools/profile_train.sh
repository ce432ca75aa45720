#!/bin/bash
# rocprofv3 kernel stats of two training steps (tools/train_step.py).  usage: profile_train.sh <tag>
set -o pipefail
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_train_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/train_step.py bm403 3 > $OUT/steps.log 2> $OUT/err.log
find $OUT -name "*kernel_stats.csv" | head -3
