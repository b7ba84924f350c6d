#!/bin/bash
# streaming GEMMs with the activation loads (1) / the output stores (2) dropped: timing only, results wrong by design
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/abl; mkdir -p $O
for v in 0 1 2; do
  if [ $v = 0 ]; then unset GCNX_LIB; else export GCNX_LIB=scripts/variants/libgcnx_abl$v.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/t$v -- python3 bench.py --workload block1m --steps 10 --warmup 3 --cpu-seconds 0 --allow-knobs --spmm-iters 2 > $O/t$v.log 2>&1
  python3 - $O/t$v $v <<'PY'
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"))[-1]
for row in csv.DictReader(open(f)):
    n = row['Name'].replace('(anonymous namespace)::','')
    if 'gemm_stream_kernel' in n or 'gemm_dw_stream' in n: print(f"abl {sys.argv[2]}  {n[:64]:64s} avg {float(row['AverageNs'])/1e3:8.1f} us")
PY
done
find $O -name "*kernel_trace.csv" -delete
