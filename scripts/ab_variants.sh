#!/bin/bash
# Same-box A/B of variant libraries (scripts/build_variant.sh NAME ...) against the product build, on one SpMM workload:
#   scripts/ab_variants.sh <block1m|powerlaw|ecoli> "base v1 v2 ... base" [spmm_bench.py args ...]
# ("base" = gcn-string_amd/gcnx/libgcnx.so; list it first AND last to see the box's drift).  Run through gpurun / scripts/gpu.sh.
cd "$GRAFT_REPO_ROOT"
WL=$1; LIST=$2; shift 2
for v in $LIST; do
  echo "== $WL $v"
  if [ $v = base ]; then L=""; else L=$PWD/scripts/variants/libgcnx_$v.so; fi
  GCNX_LIB=$L timeout -k 10 240 python scripts/spmm_bench.py --workload $WL --iters 20 --rounds 2 --slabs 0 "$@" 2>&1 | grep "^round"
done
