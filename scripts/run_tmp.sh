#!/bin/bash
# scratch: config-2 step time against the split-K chunk of gcnx_gemm_dw2
for kc in 0 10 11 12 16 23 24; do
  GCNX_DW2_KC=$kc python bench.py --allow-knobs --steps 300 --warmup 30 --cpu-seconds 0 --no-config3 > gpurun_out/kc.json 2>gpurun_out/kc.err || { tail -3 gpurun_out/kc.err; exit 1; }
  python -c "
import json; d=json.load(open('gpurun_out/kc.json')); print('kc', $kc, d['ms_per_step'], d['m1_median']['ms_per_step'])"
done
