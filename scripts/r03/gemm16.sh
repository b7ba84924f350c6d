#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/gemm16; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -x -q -m gpu -k "bf16_storage or gemm or dense or relu_bits or dx_bits" > $O/tests.log 2>&1; rc=$?; tail -6 $O/tests.log; [ $rc = 0 ] || exit $rc
for p in bf16 bf16x3 bf16; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$p -- python3 bench.py --workload block1m --prec $p --steps 20 --warmup 3 --cpu-seconds 0 > $O/trace_$p.log 2>&1
  grep -o '"ms_per_step": [0-9.]*' $O/trace_$p.log | head -1
  python3 - $O/trace_$p <<'PY'
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"))[-1]
for row in csv.DictReader(open(f)):
    n = row['Name'].replace('(anonymous namespace)::','')
    if 'gemm_' in n or 'splitk' in n: print(f"   {n[:80]:80s} calls {row['Calls']:>5s} avg {float(row['AverageNs'])/1e3:9.1f} us")
PY
done
find $O -name "*kernel_trace.csv" -delete
