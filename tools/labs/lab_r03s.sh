#!/bin/bash
# round 3, lab s: kernel durations of the training step in the SERIAL schedule (every kernel alone on the chip)
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r03lab_s; mkdir -p $out
export TMPDIR=/tmp TRAIN_STEPS=4; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $R/tools/fwd_time.py --workload bm403 --train --serial --iters 4 > $out/run.log 2>&1
S=$(find $out/stats -name "*kernel_stats.csv" | head -1); cp "$S" $out/kernel_stats_serial_train.csv
rm -rf $out/stats
head -40 $out/kernel_stats_serial_train.csv | cut -d, -f1-4
