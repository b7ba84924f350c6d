#!/bin/bash
# HBM traffic and L2 hit rate of the config-5 aggregation: column blocks (slab 0) against the r3 row gather + hub segments (cb0)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/cb_pmc; mkdir -p $O
for v in 0 cb0; do
  for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    tag=$(echo $c | cut -d' ' -f1)
    timeout -k 10 280 rocprofv3 --pmc $c --output-format csv -d $O/${v}_$tag -- python3 scripts/spmm_bench.py --workload powerlaw --iters 4 --rounds 1 --slabs $v > $O/${v}_$tag.log 2>&1 || { echo "pass $v $c failed"; tail -5 $O/${v}_$tag.log; exit 1; }
    python3 scripts/pmc_by_kernel.py $O/${v}_$tag spmm > $O/${v}_$tag.txt
    find $O/${v}_$tag -name "*counter_collection.csv" -size +3M -delete
  done
done
for f in $O/*.txt; do echo "== $f"; cat $f; done
