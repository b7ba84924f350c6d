#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for d in 0 7; do echo "== FUSED_DBG=$d"; GCNX_LIB=$GRAFT_REPO_ROOT/scripts/variants/libgcnx_tuning.so GCNX_FUSED_DBG=$d python3 scripts/fused_bench.py 2>&1 | grep -E "fwd  \(S|bwd"; done
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "head_inside" 2>&1 | tail -2
for hl in 1 1; do
GCNX_HEAD_LATE=$hl python3 bench.py --allow-knobs --steps 2000 --warmup 50 --cpu-seconds 0 --no-config3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('head_late $hl', d['ms_per_step'], d['value'], d['final_loss'])"
done
