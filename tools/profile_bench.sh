#!/bin/bash
# rocprofv3 passes over bench.py (run on the GPU box through gpurun).  usage: profile_bench.sh <tag>
# 1. kernel trace + stats of the default command (layer wavefront on)
# 2. kernel trace + stats of the wavefront-off command (the configuration roofline.avg_launch_ms is measured in)
set -o pipefail
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/default -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/default.json 2> $OUT/default.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/serial -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --serial-streams > $OUT/serial.json 2> $OUT/serial.err
echo "[profile] stats done" | tee -a $OUT/progress.log
# (HBM traffic counters: tools/pmc_traffic.sh - FETCH_SIZE and WRITE_SIZE need separate passes)
find $OUT -name "*.csv" | head -20
