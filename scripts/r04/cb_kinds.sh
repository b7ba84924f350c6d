#!/bin/bash
# time of each kind of column-block item alone (tuning build; results incomplete by design)
cd "$GRAFT_REPO_ROOT"
export GCNX_LIB=$PWD/scripts/variants/libgcnx_tuning.so
for m in 15 9 10 12; do
  echo "== spmm_cb=$m (mask $((m-8)))"
  timeout -k 10 200 python scripts/spmm_bench.py --workload powerlaw --iters 20 --rounds 2 --slabs 0 --cb $m 2>&1 | tail -2
done
