#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/gnnt; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_model.py -x -q -m gpu -k "general" > $O/tests.log 2>&1; rc=$?; tail -5 $O/tests.log; [ $rc = 0 ] || exit $rc
for p in f32 bf16x3; do
python3 bench.py --model generalgnn --prec $p --steps 50 --warmup 5 --cpu-seconds 0 > $O/b_$p.json 2> $O/b_$p.err; echo "$p $(grep -o '"ms_per_step": [0-9.]*' $O/b_$p.json | head -1)"
done
