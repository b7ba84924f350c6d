#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/t_full.log 2>&1 || { tail -30 gpurun_out/t_full.log; exit 1; }
tail -3 gpurun_out/t_full.log
for i in 1 2; do
python3 bench.py --steps 2000 --warmup 50 --cpu-seconds 0 --no-config3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('f32', d['ms_per_step'], d['value'])"
python3 bench.py --prec bf16x3 --steps 2000 --warmup 50 --cpu-seconds 0 --no-config3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('bf16x3', d['ms_per_step'], d['value'])"
done
