#!/bin/bash
# Per-kernel stats of the GeneralGNN bench line (rocprofv3 kernel trace).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/gnn; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 bench.py --model generalgnn --steps 50 --warmup 10 --cpu-seconds 0 --no-config3 > $O/t.log 2>&1
python3 scripts/kstats.py $O/t > $O/kstats.txt; tail -1 $O/t.log | head -c 300 >> $O/kstats.txt
find $O -name "*kernel_trace.csv" -delete
