#!/bin/bash
# Same-box A/B of the cache-policy variants of the aggregation kernels (scripts/build_variant.sh: ntdma, ntst, ntrows, ntall)
cd "$GRAFT_REPO_ROOT"
for wl in block1m powerlaw; do
  for v in base ntdma ntst ntrows ntall base; do
    echo "== $wl $v"
    if [ $v = base ]; then L=""; else L=$PWD/scripts/variants/libgcnx_$v.so; fi
    GCNX_LIB=$L timeout -k 10 200 python scripts/spmm_bench.py --workload $wl --iters 20 --rounds 2 --slabs 0 --cb 0 2>&1 | tail -2
  done
done
