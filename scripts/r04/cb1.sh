#!/bin/bash
# column-block kernel: tests, then config-5 aggregation A/B (cb on / off) in one process
cd "$GRAFT_REPO_ROOT"
python -m pytest tests -m gpu -x -q -k "column_blocks or hub_rows or long_rows or aggregate_and_pool or linear_head" > gpurun_out/r04_cb_tests.log 2>&1; rc=$?
tail -4 gpurun_out/r04_cb_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python scripts/spmm_bench.py --workload powerlaw --iters 20 --rounds 3 --slabs 0,cb0 > gpurun_out/r04_cb_bench.log 2>&1
tail -8 gpurun_out/r04_cb_bench.log
