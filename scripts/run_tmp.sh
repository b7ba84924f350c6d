#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/b16t; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 scripts/spmm_bench.py --workload block1m --rounds 1 --iters 5 --slabs bf16 > $O/t.log 2>&1
python3 - <<EOF2
import csv,glob
f=glob.glob("$O/t/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:6]:
    print(r["Name"][:100].ljust(100), r["Calls"], r["AverageNs"])
EOF2
find $O -name "*kernel_trace.csv" -delete
