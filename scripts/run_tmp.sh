#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for p in f32 bf16x3 bf16; do python3 bench.py --model generalgnn --prec $p --steps 50 --warmup 5 --cpu-seconds 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('generalgnn $p', round(d['ms_per_step'],4), round(d['value']))"; done
