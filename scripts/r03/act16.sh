#!/bin/bash
# bf16 storage of the GEMM-only activations: kernel tests, the model's bit-identity test, config-3 step with and without
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/act16; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -x -q -m gpu -k "bf16_storage or bf16_result_rows" > $O/tests.log 2>&1; rc=$?; tail -15 $O/tests.log; [ $rc = 0 ] || exit $rc
for v in 1 0 1 0; do
  GCNX_ACT16=$v timeout -k 10 300 python3 bench.py --workload block1m --steps 20 --warmup 3 --cpu-seconds 0 --allow-knobs > $O/bench_$v.json 2> $O/bench_$v.err || { tail -5 $O/bench_$v.err; exit 1; }
  python3 -c "
import json
for l in open('$O/bench_$v.json'):
    if l.startswith('{'):
        r=json.loads(l); print('act16=$v', round(r['ms_per_step'],4), 'ms/step', round(r['value']), 'graphs/s loss', r['final_loss'])
"
done
