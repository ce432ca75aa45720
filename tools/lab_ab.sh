#!/bin/bash
# A/B of two builds on ONE box: the product library against lib/libmatgcn_prev.so, training-step timing, twice each
for i in 1 2; do
  echo "== product"; python tools/host_enqueue_time.py bm403 6 2>&1 | tail -2
  echo "== prev"; MATGCN_LIB=$PWD/multistgraph_amd/lib/libmatgcn_prev.so python tools/host_enqueue_time.py bm403 6 2>&1 | tail -2
done
