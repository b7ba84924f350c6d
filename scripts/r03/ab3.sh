#!/bin/bash
# same-box A/B/C: HEAD, + whole-row pool, + bit-image touch prefetch
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/ab3; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "pool or fold or bf16_result_rows" > $O/tests.log 2>&1; rc=$?; tail -5 $O/tests.log; [ $rc = 0 ] || exit $rc
for rep in 1 2 3; do
  for lib in head v1 new; do
    if [ $lib = new ]; then unset GCNX_LIB; else export GCNX_LIB=scripts/variants/libgcnx_$lib.so; fi
    timeout -k 10 300 python3 bench.py --workload block1m --steps 20 --warmup 3 --cpu-seconds 0 --allow-knobs > $O/${lib}.json 2> $O/${lib}.err || { tail -3 $O/${lib}.err; exit 1; }
    echo "$lib $(grep -o '"ms_per_step": [0-9.]*' $O/${lib}.json | head -1)"
  done
done
