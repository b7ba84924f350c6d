#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/act16; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --workload block1m --steps 20 --warmup 3 --cpu-seconds 0 > $O/trace.log 2>&1
find $O -name "*kernel_trace.csv" -delete
python3 - <<'PY'
import csv, glob
f = sorted(glob.glob("gpurun_out/act16/trace/*/*kernel_stats.csv"))[-1]
for i, row in enumerate(csv.DictReader(open(f))):
    if i < 16:
        print(f"{row['Name'].replace('(anonymous namespace)::','')[:90]:90s} calls {row['Calls']:>5s} avg {float(row['AverageNs'])/1e3:9.1f} us {row['Percentage']:>6s}%")
PY
