#!/bin/bash
# same-box A/B of two library builds: per-kernel SpMM times at config 3.  usage: ab.sh OUTDIR "dbg values" lib1 lib2 ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; mkdir -p $O; shift
DBGS=$1; shift
for rep in 1 2; do for lib in "$@"; do for d in $DBGS; do
  tag=$(basename $lib .so)_${d}_$rep
  GCNX_LIB=$GRAFT_REPO_ROOT/$lib GCNX_SPMM_DBG=$d rocprofv3 --kernel-trace --stats --output-format csv -d $O/$tag -- python3 scripts/spmm_bench.py --workload block1m --iters 10 --rounds 1 --slabs 0 > $O/$tag.log 2>&1
  echo "$tag: $(grep -h 'round 0' $O/$tag.log | sed 's/.*slab *0: *//;s/GB.*//') | $(python3 scripts/kstats.py $O/$tag spmm)" | tee -a $O/ab.txt
done; done; done
find $O -name "*kernel_trace.csv" -delete
