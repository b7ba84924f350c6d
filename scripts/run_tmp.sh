#!/bin/bash
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/t.log 2>&1 || { tail -30 gpurun_out/t.log; exit 1; }
tail -2 gpurun_out/t.log
python bench.py --model generalgnn --steps 50 --warmup 5 > gpurun_out/bg.json 2>gpurun_out/bg.err && python -c "
import json; d=json.load(open('gpurun_out/bg.json')); print('generalgnn', d['ms_per_step'], d['value'])"
python scripts/gemm_bench.py --n 1000000 --shapes 256x256 --prec f32 --iters 5
python scripts/gemm_bench.py --n 22576 --shapes 256x256,16x256 --prec f32 --iters 20
GCNX_GEMM_STREAM=0 python scripts/gemm_bench.py --n 22576 --shapes 256x256,16x256 --prec f32 --iters 20
