#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03j; mkdir -p $O
for v in "" "GCNX_SPMM_CAP1=0" "GCNX_SPMM_SG=1" "GCNX_SPMM_SG=2" "GCNX_SPMM_SG=4" "GCNX_SPMM_CAP1=0 GCNX_SPMM_SG=2" "GCNX_SPMM_CAP1=0 GCNX_SPMM_SG=4" "GCNX_SPMM_CAP1=400" "GCNX_SPMM_TALL_RPC=32"; do
  echo "== $v: $(env $v python3 scripts/spmm_bench.py --workload block1m --shard 0,8 --iters 50 --rounds 2 --slabs 0 2>&1 | grep 'round' | sed 's/GB.*//' | tr '\n' ' ')" | tee -a $O/shard_spmm.txt
done
