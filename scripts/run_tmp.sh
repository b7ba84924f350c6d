#!/bin/bash
for sl in 64 128 192 256 384; do echo "== slices $sl"; GCNX_DW_SLICES=$sl python scripts/fused_bench.py | grep dw2; done
GCNX_GEMM_STREAM=0 python scripts/fused_bench.py | grep dw2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tr_f -- python3 scripts/fused_bench.py > gpurun_out/tr_f.log 2>&1; find gpurun_out/tr_f -name "*kernel_trace.csv" -delete; python3 - <<EOF2
import csv,glob,os
f=sorted(glob.glob("gpurun_out/tr_f/*/*kernel_stats.csv"), key=os.path.getmtime)[-1]
for r in list(csv.DictReader(open(f)))[:10]:
    print(r["Name"][:90].ljust(90), r["Calls"], r["AverageNs"])
EOF2
