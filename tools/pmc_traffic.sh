#!/bin/bash
# HBM-traffic PMC passes over bench.py (wavefront off, 1+1 forwards).  FETCH_SIZE and WRITE_SIZE need separate
# passes on gfx950 (TCC has 4 slots: FETCH_SIZE takes 3, WRITE_SIZE 2); counters only, no tracing domains.
set -o pipefail
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  echo "[pmc] $C" | tee -a $OUT/progress.log
  rocprofv3 --pmc $C --output-format csv -d $OUT/$C -- python3 $R/tools/one_forward.py > $OUT/$C.log 2>&1
  echo "[pmc] $C done rc=$?" | tee -a $OUT/progress.log
done
find $OUT -name "*counter_collection.csv"
