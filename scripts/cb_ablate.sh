#!/bin/bash
# Ablation of the column-block kernel (config 5) on a TUNING build (scripts/build_variant.sh tuning "-DGCNX_TUNING" spmm):
#   CBS="24 19 31"  kinds launched: 16 + mask (8 short rows, 2 rows of 33..512 entries, 1 hub rows); results incomplete by design
#   DBGS="0 1 4 5"  GCNX_CB_DBG bits: 1 no stores, 4 no gathers, 8 plain (not nt) stores, 16 launch + item only,
#                   32 + row records only, 64 no entry loads; results wrong by design
cd "$GRAFT_REPO_ROOT"
export GCNX_LIB=$PWD/scripts/variants/libgcnx_tuning.so
for cb in ${CBS:-24 19 31}; do
for d in ${DBGS:-0 1 4 5}; do
  echo "== cb=$cb dbg=$d"
  GCNX_CB_DBG=$d timeout -k 10 200 python scripts/spmm_bench.py --workload powerlaw --iters 20 --rounds 1 --slabs 0 --cb $cb 2>&1 | tail -1
done
done
