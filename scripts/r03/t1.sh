#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/t1
timeout -k 10 900 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -x -q -m gpu -k "relu_bits_pool or pooled_layer or bf16_storage or config3 or rccl or world_size_2" > gpurun_out/t1/tests.log 2>&1; rc=$?; tail -15 gpurun_out/t1/tests.log; [ $rc = 0 ] || exit $rc
for v in 1 0 1 0; do
  GCNX_POOL_IN_SPMM=$v timeout -k 10 300 python3 bench.py --workload block1m --steps 20 --warmup 3 --cpu-seconds 0 --allow-knobs > gpurun_out/t1/b_$v.json 2> gpurun_out/t1/b_$v.err || { tail -3 gpurun_out/t1/b_$v.err; exit 1; }
  echo "pool_in_spmm=$v $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/t1/b_$v.json | head -1)"
done
