#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03m; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_model.py -x -q -m gpu -k "general_gnn" > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -3 $O/tests.log
for p in f32 bf16x3 bf16; do
  python3 bench.py --model generalgnn --prec $p --steps 100 --warmup 10 > $O/gnn_$p.json 2> $O/gnn_$p.err || tail -5 $O/gnn_$p.err
  python3 -c "import json; r=json.load(open('$O/gnn_$p.json')); print('$p', r['ms_per_step'], r['value'])"
done
