#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for cfg in "0 0" "1 0" "1 224" "1 208" "1 192" "1 176" "1 160" "0 192" "0 0"; do
  set -- $cfg
  echo "== conc=$1 tile_wgs=$2"
  timeout -k 10 240 python scripts/spmm_bench.py --workload block1m --iters 20 --rounds 2 --slabs 0 --conc $1 --tile-wgs $2 2>&1 | grep "^round"
done
