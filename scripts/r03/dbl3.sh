#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03p; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "spmm or pool_bwd" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
export GCNX_LIB=$GRAFT_REPO_ROOT/scripts/variants/libgcnx_tuning.so
for rep in 1 2; do for d in 0 32 16; do
  GCNX_SPMM_DBG=$d rocprofv3 --kernel-trace --stats --output-format csv -d $O/u_${d}_$rep -- python3 scripts/spmm_bench.py --workload block1m --iters 10 --rounds 1 --slabs 0 > $O/u_${d}_$rep.log 2>&1
  echo "dbg=$d: $(python3 scripts/kstats.py $O/u_${d}_$rep spmm_duo) $(grep -h round $O/u_${d}_$rep.log | sed 's/GB.*//')"
done; done
find $O -name "*kernel_trace.csv" -delete
