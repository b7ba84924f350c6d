#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu 2>&1 | tail -8
for p in bf16 f32; do python3 bench.py --workload block1m --prec $p --steps 20 --warmup 3 --cpu-seconds 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('block1m $p', round(d['ms_per_step'],4), round(d['value']), d['final_loss'])"; done
python3 bench.py --workload powerlaw --steps 20 --warmup 3 --cpu-seconds 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('powerlaw', round(d['ms_per_step'],4), round(d['value']), d['final_loss'])"
GCNX_S_ORDER=0 python3 bench.py --allow-knobs --workload block1m --steps 20 --warmup 3 --cpu-seconds 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('block1m bf16 s_order=0', round(d['ms_per_step'],4), d['final_loss'])"
