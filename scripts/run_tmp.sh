#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 scripts/fused_bench.py 2>&1 | grep dw2
timeout -k 10 400 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "gemm or dense_bwd or fused_backward or head_inside" 2>&1 | tail -3
for i in 1 2; do python3 bench.py --steps 2000 --warmup 50 --cpu-seconds 0 --no-config3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('ecoli', d['ms_per_step'], d['value'], d['final_loss'])"; done
python3 bench.py --model generalgnn --steps 50 --warmup 5 --cpu-seconds 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('generalgnn', d['ms_per_step'], d['value'])"
python3 scripts/gemm_bench.py --n 1000000 --shapes 256x256 --prec f32 --iters 5 2>&1 | tail -1
