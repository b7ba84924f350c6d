#!/bin/bash
# does the relative placement of the gathered matrix and the output move the config-3 SpMM time?
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 400 python3 scripts/spmm_bench.py --workload block1m --rounds 2 --iters 20 --slabs 0 --pads 0,2,6,34,130,258,1,0.5,0.0625,977,0 > gpurun_out/pads.txt 2>&1
tail -16 gpurun_out/pads.txt
