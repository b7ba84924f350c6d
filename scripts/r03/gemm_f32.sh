#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 300 python3 scripts/gemm_bench.py --n 22576 --prec f32 --iters 30 --shapes 16x256,256x256,512x256,768x256,1024x256 > gpurun_out/gemm_f32_22k.txt 2>&1; cat gpurun_out/gemm_f32_22k.txt
timeout -k 10 300 python3 scripts/gemm_bench.py --n 22576 --prec bf16x3 --iters 30 --shapes 256x256,512x256,1024x256 >> gpurun_out/gemm_f32_22k.txt 2>&1; tail -3 gpurun_out/gemm_f32_22k.txt
