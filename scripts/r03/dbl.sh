#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03p; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "spmm or pool_bwd" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for rep in 1 2; do for lib in scripts/variants/libgcnx_r2.so gcn-string_amd/gcnx/libgcnx.so; do
  echo "$lib: $(GCNX_LIB=$GRAFT_REPO_ROOT/$lib python3 scripts/spmm_bench.py --workload block1m --iters 20 --rounds 1 --slabs 0 2>&1 | grep round)"
done; done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -- python3 scripts/spmm_bench.py --workload block1m --iters 10 --rounds 1 --slabs 0 > $O/tr.log 2>&1
echo "kernels: $(python3 scripts/kstats.py $O/tr spmm)"
find $O -name "*kernel_trace.csv" -delete
