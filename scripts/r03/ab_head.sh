#!/bin/bash
# same-box A/B: the committed library (scripts/variants/libgcnx_head.so) against the working tree's, config 3 per precision
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/abhead; mkdir -p $O
for rep in 1 2; do
for p in bf16x3 bf16 f32; do
  for lib in head new; do
    if [ $lib = head ]; then export GCNX_LIB=scripts/variants/libgcnx_head.so; else unset GCNX_LIB; fi
    timeout -k 10 300 python3 bench.py --workload block1m --prec $p --steps 20 --warmup 3 --cpu-seconds 0 --allow-knobs > $O/${p}_${lib}.json 2> $O/${p}_${lib}.err || { tail -3 $O/${p}_${lib}.err; exit 1; }
    echo "$p $lib $(grep -o '"ms_per_step": [0-9.]*' $O/${p}_${lib}.json | head -1)"
  done
done
done
