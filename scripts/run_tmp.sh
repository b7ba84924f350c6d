#!/bin/bash
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "fused" > gpurun_out/t.log 2>&1 || { tail -20 gpurun_out/t.log; exit 1; }
tail -2 gpurun_out/t.log
cp gcn-string_amd/gcnx/libgcnx.so /tmp/libgcnx_release.so
cp scripts/variants/libgcnx_tuning.so gcn-string_amd/gcnx/libgcnx.so
for d in 0 1; do echo "== dbg $d"; GCNX_FUSED_DBG=$d python scripts/fused_bench.py || exit 1; done
cp /tmp/libgcnx_release.so gcn-string_amd/gcnx/libgcnx.so
python bench.py --steps 300 --warmup 30 --cpu-seconds 0 --no-config3 > gpurun_out/b2.json 2>gpurun_out/b2.err && python -c "
import json; d=json.load(open('gpurun_out/b2.json')); print(d['ms_per_step'], d['m1_median']['ms_per_step'])"
