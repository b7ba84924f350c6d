#!/bin/bash
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -x -q -k "fused or golden or train or rccl" > gpurun_out/t.log 2>&1 || { tail -30 gpurun_out/t.log; exit 1; }
tail -2 gpurun_out/t.log
python scripts/fused_bench.py
for p in f32 bf16x3; do
python bench.py --prec $p --steps 300 --warmup 30 --cpu-seconds 0 --no-config3 > gpurun_out/b2.json 2>gpurun_out/b2.err && python -c "
import json; d=json.load(open('gpurun_out/b2.json')); print('$p', d['ms_per_step'], d['m1_median']['ms_per_step'], d['final_loss'], d.get('roofline_step_kernel',{}).get('avg_launch_us'))"
done
