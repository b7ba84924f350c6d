#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for v in o4_c6 o2_c8 o2_c6 o1_c8; do
  echo "== $v"
  GCNX_LIB=$PWD/scripts/variants/libgcnx_cb_$v.so timeout -k 10 200 python scripts/spmm_bench.py --workload powerlaw --iters 20 --rounds 2 --slabs 0,cb0 2>&1 | tail -4
done
