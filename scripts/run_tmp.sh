cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "spmm" 2>&1 | tail -5
for k in pipe tile; do echo "== $k"; GCNX_SPMM_KERNEL=$k timeout -k 10 200 python scripts/spmm_bench.py --workload block1m --rounds 2 --iters 10 --slabs 0 2>&1 | tail -2; done
export GCNX_LIB=$GRAFT_REPO_ROOT/scripts/variants/lib_tune.so
GCNX_SPMM_STAMPS=1 GCNX_SPMM_KERNEL=pipe timeout -k 10 200 python scripts/spmm_bench.py --workload block1m --rounds 1 --iters 2 --slabs 0 2>&1 | tail -5
for d in 1 2 4 7; do echo "== pipe DBG=$d"; GCNX_SPMM_DBG=$d GCNX_SPMM_KERNEL=pipe timeout -k 10 200 python scripts/spmm_bench.py --workload block1m --rounds 1 --iters 10 --slabs 0 2>&1 | tail -1; done
