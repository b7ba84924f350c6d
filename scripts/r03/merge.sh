#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/merge; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -x -q -m gpu -k "bf16_storage or gemm or dense or relu_bits or dx_bits or config3" > $O/tests.log 2>&1; rc=$?; tail -6 $O/tests.log; [ $rc = 0 ] || exit $rc
for rep in 1 2; do
  for lib in head new; do
    if [ $lib = head ]; then export GCNX_LIB=scripts/variants/libgcnx_head.so; else unset GCNX_LIB; fi
    [ $lib = head ] && continue
    timeout -k 10 300 python3 bench.py --workload block1m --steps 20 --warmup 3 --cpu-seconds 0 --allow-knobs > $O/full_${lib}.json 2> $O/full_${lib}.err || { tail -3 $O/full_${lib}.err; exit 1; }
    echo "full $lib $(grep -o '"ms_per_step": [0-9.]*' $O/full_${lib}.json | head -1)"
    timeout -k 10 300 python3 bench.py --workload block1m --steps 30 --warmup 3 --cpu-seconds 0 --no-config3 --spmm-iters 2 --emulate-rank 4 --of 8 --allow-knobs > $O/shard_${lib}.json 2> $O/shard_${lib}.err || { tail -3 $O/shard_${lib}.err; exit 1; }
    echo "shard4/8 $lib $(grep -o '"ms_per_step": [0-9.]*' $O/shard_${lib}.json | head -1)"
  done
done
