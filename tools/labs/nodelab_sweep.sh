#!/bin/bash
# runs every tools/nodelab_* variant, keeping the product-kernel lines
mkdir -p gpurun_out
for b in tools/nodelab_base tools/nodelab_10_10_6 tools/nodelab_10_8_4 tools/nodelab_8_8_5 tools/nodelab_10_6_4; do
  echo "== $b" >> gpurun_out/nodelab_sweep.log
  timeout -k 10 120 $b 2>&1 | grep -E "gate16|update16|px16" >> gpurun_out/nodelab_sweep.log || exit 1
done
