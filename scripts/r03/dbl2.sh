#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03p; mkdir -p $O
export GCNX_LIB=$GRAFT_REPO_ROOT/scripts/variants/libgcnx_tuning.so
for rep in 1 2; do for d in 0 16 2 18; do
  GCNX_SPMM_DBG=$d rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_${d}_$rep -- python3 scripts/spmm_bench.py --workload block1m --iters 10 --rounds 1 --slabs 0 > $O/t_${d}_$rep.log 2>&1
  echo "dbg=$d: $(python3 scripts/kstats.py $O/t_${d}_$rep spmm_duo)"
done; done
find $O -name "*kernel_trace.csv" -delete
