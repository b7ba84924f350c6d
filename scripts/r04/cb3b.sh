#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export GCNX_LIB=$PWD/scripts/variants/libgcnx_tuning.so
for cb in ${CBS:-24}; do
for d in ${DBGS:-16 32 5}; do
  echo "== cb=$cb dbg=$d"
  GCNX_CB_DBG=$d timeout -k 10 200 python scripts/spmm_bench.py --workload powerlaw --iters 20 --rounds 1 --slabs 0 --cb $cb 2>&1 | tail -1
done
done
