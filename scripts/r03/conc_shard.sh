#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/conc; mkdir -p $O
for rep in 1 2; do
for v in 0 1; do
  GCNX_SPMM_CONC=$v timeout -k 10 300 python3 bench.py --workload block1m --steps 40 --warmup 3 --cpu-seconds 0 --no-config3 --spmm-iters 2 --emulate-rank 4 --of 8 --allow-knobs > $O/s_$v.json 2> $O/s_$v.err || { tail -3 $O/s_$v.err; exit 1; }
  echo "shard4/8 conc=$v $(grep -o '"ms_per_step": [0-9.]*' $O/s_$v.json | head -1)"
done
done
