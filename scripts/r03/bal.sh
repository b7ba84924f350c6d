#!/bin/bash
# balanced deal of the tile graphs (GCNX_SPMM_BAL): spmm tests bit-exact, then A/B of cost models at config 3, alternating
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/bal; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "spmm" > $O/tests.log 2>&1; rc=$?; tail -3 $O/tests.log; [ $rc = 0 ] || exit $rc
for rep in 1 2; do for b in 0 1 100024 24 10060 25024 10000; do
  tag=b${b}_$rep
  GCNX_SPMM_BAL=$b rocprofv3 --kernel-trace --stats --output-format csv -d $O/$tag -- python3 scripts/spmm_bench.py --workload block1m --iters 20 --rounds 1 --slabs 0 > $O/$tag.log 2>&1
  echo "$tag: $(grep -h 'round 0' $O/$tag.log | sed 's/.*slab *0: *//;s/GB.*//') | $(python3 scripts/kstats.py $O/$tag spmm)" | tee -a $O/ab.txt
done; done
find $O -name "*kernel_trace.csv" -delete
