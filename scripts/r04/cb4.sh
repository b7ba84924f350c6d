#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_kernels.py -x -q -k "column_blocks" 2>&1 | tail -2
for v in base:1 base:2 hub64:1 hub64hu8:1 hub128hu8:1 hub512hu8:1 base:1; do
  lib=${v%%:*}; cb=${v##*:}
  echo "== $lib cb=$cb"
  if [ $lib = base ]; then L=""; else L=$PWD/scripts/variants/libgcnx_$lib.so; fi
  GCNX_LIB=$L timeout -k 10 200 python scripts/spmm_bench.py --workload powerlaw --iters 20 --rounds 2 --slabs 0 --cb $cb 2>&1 | tail -2
done
