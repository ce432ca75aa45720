#!/bin/bash
# All the profiler passes a bench line's roofline object refers to (run on the GPU box through gpurun).
#   usage: profile_run.sh <tag>     -> gpurun_out/prof_<tag>/{stats_serial,stats_default,FETCH_SIZE,WRITE_SIZE,MFMA}
# 1/2  rocprofv3 --kernel-trace --stats of bench.py with the wavefront off (the configuration avg_launch_ms is measured
#      in) and of the default command
# 3-5  PMC passes over two serial-schedule forwards (counters only, no tracing domains; FETCH_SIZE and WRITE_SIZE need
#      separate passes on gfx950), summarised per kernel by tools/summarise_pmc.py
set -o pipefail
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_serial -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train-step --no-batch16 --serial-streams > $OUT/bench_serial.json 2> $OUT/bench_serial.err
echo "[profile] serial stats rc=$?" | tee -a $OUT/progress.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_default -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train-step --no-batch16 > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "[profile] default stats rc=$?" | tee -a $OUT/progress.log
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/$C -- python3 $R/tools/one_forward.py > $OUT/$C.log 2>&1
  echo "[pmc] $C rc=$?" | tee -a $OUT/progress.log
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/MFMA -- python3 $R/tools/one_forward.py > $OUT/MFMA.log 2>&1
echo "[pmc] MFMA rc=$?" | tee -a $OUT/progress.log
cd $R
F=$(find $OUT/FETCH_SIZE -name "*counter_collection.csv" | head -1)
W=$(find $OUT/WRITE_SIZE -name "*counter_collection.csv" | head -1)
M=$(find $OUT/MFMA -name "*counter_collection.csv" | head -1)
python3 tools/summarise_pmc.py "$F" "$W" "$M" $OUT/pmc_kernels.json $TAG | tee $OUT/pmc_kernels.txt
for m in serial default; do
  S=$(find $OUT/stats_$m -name "*kernel_stats.csv" | head -1)
  [ -n "$S" ] && cp "$S" $OUT/kernel_stats_$m.csv
done
ls $OUT
