#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/gnnf32; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --model generalgnn --prec f32 --steps 50 --warmup 5 --cpu-seconds 0 > $O/trace.log 2>&1
find $O -name "*kernel_trace.csv" -delete
python3 - <<'PY'
import csv, glob
f = sorted(glob.glob("gpurun_out/gnnf32/trace/*/*kernel_stats.csv"))[-1]
rows = list(csv.DictReader(open(f)))
steps = None
for r in rows:
    if 'softmax_cce' in r['Name']: steps = int(r['Calls'])
tot = 0
for r in rows[:30]:
    per = float(r['TotalDurationNs'])/1e3/steps; tot += per
    print(f"{r['Name'].replace('(anonymous namespace)::','')[:70]:70s} calls/step {int(r['Calls'])/steps:5.1f} avg {float(r['AverageNs'])/1e3:7.1f} us  per step {per:7.1f}")
print('sum', tot, 'steps', steps)
PY
