#!/bin/bash
for p in bf16x3 bf16 bf16x3; do
python bench.py --workload block1m --prec $p --steps 20 --warmup 3 --cpu-seconds 0 > gpurun_out/b1.json 2>gpurun_out/b1.err && python -c "
import json; d=json.load(open('gpurun_out/b1.json')); print('$p', d['ms_per_step'], d['m1_median']['ms_per_step'], d['final_loss'])"
done
