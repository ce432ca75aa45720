#!/bin/bash
# rocprofv3 passes over bench.py (run on the GPU box through gpurun).  usage: profile_bench.sh <tag>
# 1. kernel trace + stats of the default command (layer wavefront on)
# 2. kernel trace + stats of the wavefront-off command (the configuration roofline.avg_launch_ms is measured in)
# 3. PMC pass (counters only, no tracing domains) for HBM traffic of the graph-mix kernel
set -o pipefail
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/default -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/default.json 2> $OUT/default.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/serial -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --serial-streams > $OUT/serial.json 2> $OUT/serial.err
rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --output-format csv -d $OUT/pmc_hbm -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --serial-streams > $OUT/pmc.json 2> $OUT/pmc.err
find $OUT -name "*.csv" | head -20
