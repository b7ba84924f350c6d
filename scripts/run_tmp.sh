cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "spmm" 2>&1 | tail -5
for k in auto tile pipe; do echo "== GCNX_SPMM_KERNEL=$k"; GCNX_SPMM_KERNEL=$k timeout -k 10 200 python scripts/spmm_bench.py --workload block1m --rounds 2 --iters 10 --slabs 0 2>&1 | tail -2; done
