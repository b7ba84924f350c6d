#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03c; mkdir -p $O
export GCNX_LIB=$GRAFT_REPO_ROOT/scripts/variants/libgcnx_tuning.so
for d in 0 1 2 3 4 6 14 15; do
  GCNX_SPMM_DBG=$d rocprofv3 --kernel-trace --stats --output-format csv -d $O/abl_$d -- python3 scripts/spmm_bench.py --workload block1m --iters 10 --rounds 1 --slabs 0 > $O/abl_$d.log 2>&1
  echo "tiers dbg=$d: $(python3 scripts/kstats.py $O/abl_$d spmm)" | tee -a $O/ablation.txt
done
find $O -name "*kernel_trace.csv" -delete
