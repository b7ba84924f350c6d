#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_kernels.py -x -q -k "column_blocks" 2>&1 | tail -3
echo "== product: default path (cb0) vs column blocks"
timeout -k 10 200 python scripts/spmm_bench.py --workload powerlaw --iters 20 --rounds 2 --slabs 0,cb0 2>&1 | tail -4
export GCNX_LIB=$PWD/scripts/variants/libgcnx_tuning.so
for cb in 24 28 19; do
for d in ${DBGS:-0 1 4 5 8}; do
  echo "== cb=$cb dbg=$d"
  GCNX_CB_DBG=$d timeout -k 10 200 python scripts/spmm_bench.py --workload powerlaw --iters 20 --rounds 1 --slabs 0 --cb $cb 2>&1 | tail -1
done
done
