#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03j; mkdir -p $O
for sh in "" "--shard 0,2" "--shard 0,4" "--shard 0,8"; do
for v in "" "GCNX_SPMM_CAP1=0" "GCNX_SPMM_CAP1=0 GCNX_SPMM_SG=4" "GCNX_SPMM_CAP1=0 GCNX_SPMM_SG=8"; do
  echo "== [$sh] $v: $(env $v python3 scripts/spmm_bench.py --workload block1m $sh --iters 30 --rounds 2 --slabs 0 2>&1 | grep 'round' | sed 's/GB.*//;s/round . slab *0://' | tr '\n' ' ')" | tee -a $O/shard_spmm2.txt
done; done
