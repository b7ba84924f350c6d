#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_kernels.py -x -q 2>&1 | tail -3
python -m pytest tests/test_gpu_model.py -x -q -k "config3_and_config5 or powerlaw or power_law" 2>&1 | tail -3
echo "== product: column blocks vs the r3 path (cb0)"
timeout -k 10 200 python scripts/spmm_bench.py --workload powerlaw --iters 20 --rounds 3 --slabs 0,cb0 2>&1 | tail -6
python bench.py --workload powerlaw --steps 20 --warmup 3 --cpu-seconds 0 > gpurun_out/bench_powerlaw_cb.json 2> gpurun_out/bench_powerlaw_cb.err; tail -c 1200 gpurun_out/bench_powerlaw_cb.json
