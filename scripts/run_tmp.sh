#!/bin/bash
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -x -q -k "fold or config3 or config5 or spmm" > gpurun_out/t.log 2>&1 || { tail -30 gpurun_out/t.log; exit 1; }
tail -2 gpurun_out/t.log
python scripts/spmm_bench.py --slabs 0 --iters 20 --rounds 2
for p in bf16 bf16x3; do
python bench.py --workload block1m --prec $p --steps 20 --warmup 3 --cpu-seconds 0 > gpurun_out/b1.json 2>gpurun_out/b1.err && python -c "
import json; d=json.load(open('gpurun_out/b1.json')); print('$p', d['ms_per_step'], d['m1_median']['ms_per_step'], d['final_loss'], d['roofline']['avg_launch_us'])"
done
