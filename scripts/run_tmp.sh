#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for hl in 1 0 1; do
GCNX_HEAD_LATE=$hl python3 bench.py --allow-knobs --steps 2000 --warmup 50 --cpu-seconds 0 --no-config3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('head_late $hl', d['ms_per_step'], d['value'], d['final_loss'])"
done
python3 bench.py --prec bf16x3 --steps 2000 --warmup 50 --cpu-seconds 0 --no-config3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('bf16x3', d['ms_per_step'], d['value'])"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tr_h -- python3 bench.py --steps 200 --warmup 20 --cpu-seconds 0 --no-config3 > gpurun_out/tr_h.log 2>&1
find gpurun_out/tr_h -name "*kernel_trace.csv" -delete
python3 - <<EOF2
import csv,glob,os
f=sorted(glob.glob("gpurun_out/tr_h/*/*kernel_stats.csv"), key=os.path.getmtime)[-1]
for r in list(csv.DictReader(open(f)))[:8]:
    print(r["Name"][:90].ljust(90), r["Calls"], r["AverageNs"], r["Percentage"])
EOF2
