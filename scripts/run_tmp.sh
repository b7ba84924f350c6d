#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu 2>&1 | tail -4
python3 bench.py --workload block1m --steps 20 --warmup 3 --cpu-seconds 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('block1m', round(d['ms_per_step'],4), round(d['value']))"
