#!/bin/bash
# rocprofv3 kernel stats of three training steps on the SERIAL schedule (every kernel alone on the chip): the kernels' own durations
#   usage: prof_train_serial.sh <tag> [library] [workload] [batch]
set -o pipefail
TAG=${1:-r04}
R=$GRAFT_REPO_ROOT
[ -n "$2" ] && export MATGCN_LIB=$R/$2
W=${3:-bm403}
BATCH=${4:-}
OUT=$R/gpurun_out/prof_train_serial_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/train_step.py $W 3 serial $BATCH > $OUT/steps.log 2> $OUT/err.log || { tail -5 $OUT/err.log; exit 1; }
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
cp $f $OUT/kernel_stats.csv
python3 - $OUT/kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time of 3 steps %.2f ms" % (tot / 1e6))
for r in rows[:28]:
    print("%-70s calls %5s  avg %8.1f us  total/step %7.3f ms" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 3e6))
PY
tail -3 $OUT/steps.log
