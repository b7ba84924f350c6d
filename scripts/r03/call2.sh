#!/bin/bash
# r03 call 2: degree-ordered tile kernel -- correctness (spmm + pool_bwd tests), then timing per kernel
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03b; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "spmm or pool_bwd or graph_prep" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -3 $O/tests.log
python3 scripts/spmm_bench.py --workload block1m --iters 20 --rounds 2 --slabs 0 > $O/spmm.txt 2>&1 || { tail -5 $O/spmm.txt; exit 1; }
cat $O/spmm.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -- python3 scripts/spmm_bench.py --workload block1m --iters 10 --rounds 1 --slabs 0 > $O/tr.log 2>&1
echo "r3 kernels: $(python3 scripts/kstats.py $O/tr spmm)"
GCNX_SPMM_TALL_RPC=32 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr32 -- python3 scripts/spmm_bench.py --workload block1m --iters 10 --rounds 1 --slabs 0 > $O/tr32.log 2>&1
echo "tall rpc 32: $(python3 scripts/kstats.py $O/tr32 spmm)"
find $O -name "*kernel_trace.csv" -delete
