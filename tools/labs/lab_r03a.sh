#!/bin/bash
# round 3, lab a: CU-mask probe + XCD-hierarchical barrier lab (outputs under gpurun_out/r03lab_a/)
set -o pipefail
out=gpurun_out/r03lab_a; mkdir -p $out
timeout -k 10 120 tools/labs/cumask_probe > $out/cumask_probe.log 2>&1; echo "cumask_probe rc=$?" >> $out/cumask_probe.log
timeout -k 10 300 tools/labs/xcdbarrier_lab > $out/xcdbarrier_lab.log 2>&1; echo "xcdbarrier_lab rc=$?" >> $out/xcdbarrier_lab.log
cat $out/cumask_probe.log $out/xcdbarrier_lab.log
