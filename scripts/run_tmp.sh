#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in 0 1 0 1; do echo "== conc $c"; python3 scripts/spmm_bench.py --workload block1m --rounds 2 --iters 20 --conc $c 2>&1 | tail -3; done
python3 scripts/spmm_bench.py --workload powerlaw --rounds 2 --iters 10 --conc 0 2>&1 | tail -2
python3 scripts/spmm_bench.py --workload powerlaw --rounds 2 --iters 10 --conc 1 2>&1 | tail -2
