#!/bin/bash
# A/B of builds on ONE box: the product library against lab variants lib/libmatgcn_<tag>.so, training-step timing
for tag in product "$@" product; do
  if [ $tag = product ]; then unset MATGCN_LIB; else export MATGCN_LIB=$PWD/multistgraph_amd/lib/libmatgcn_$tag.so; fi
  echo "== $tag"; python tools/host_enqueue_time.py bm403 6 2>&1 | tail -2 | sed 's/host enqueue.*device: //'
done
