#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "bf16_features or bf16_long" 2>&1 | tail -15
python3 bench.py --steps 200 --warmup 20 --cpu-seconds 0 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step']); print(d['roofline_config3']['avg_launch_us'], d['roofline_config3']['frac']); print(d['roofline_config3_bf16'])"
