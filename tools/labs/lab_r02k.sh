#!/bin/bash
# what bounds k_wgrad_node: kernel time with the atomics, the MFMAs or the global loads taken out (lab builds)
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r02k
export TMPDIR=/tmp
cd /tmp
for v in base noatomic nomfma noload; do
  OUT=$R/gpurun_out/r02k/$v
  mkdir -p $OUT
  if [ $v = base ]; then unset MATGCN_LIB; else export MATGCN_LIB=$R/multistgraph_amd/lib/libmatgcn_wgn_$v.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/train_step.py bm403 3 serial > $OUT/steps.log 2> $OUT/err.log || { echo "$v failed"; tail -3 $OUT/err.log; exit 1; }
  f=$(find $OUT -name "*kernel_stats.csv" | head -1)
  echo "== $v"; grep -E "k_wgrad_node|k_chain_node<false, 192>" $f | awk -F, '{print $1, $2, $4}'
done
