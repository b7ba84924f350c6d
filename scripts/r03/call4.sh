#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03d; mkdir -p $O
export GCNX_LIB=$GRAFT_REPO_ROOT/scripts/variants/libgcnx_tuning.so
for w in 0 1 2; do for d in 0 2 3; do
  GCNX_SPMM_SORT_WIN=$w GCNX_SPMM_DBG=$d rocprofv3 --kernel-trace --stats --output-format csv -d $O/abl_${w}_$d -- python3 scripts/spmm_bench.py --workload block1m --iters 10 --rounds 1 --slabs 0 > $O/abl_${w}_$d.log 2>&1
  echo "win=$w dbg=$d: $(python3 scripts/kstats.py $O/abl_${w}_$d spmm_duo)" | tee -a $O/ablation.txt
done; done
find $O -name "*kernel_trace.csv" -delete
