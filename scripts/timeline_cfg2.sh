#!/bin/bash
# Kernel timeline of one config-2 training step (rocprofv3 kernel trace of the bench command).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/tl; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 bench.py --steps 200 --warmup 20 --cpu-seconds 0 --no-config3 --no-generalgnn > $O/t.log 2>&1
python3 scripts/ktimeline.py $O/t > $O/timeline.txt; python3 scripts/kstats.py $O/t >> $O/timeline.txt
find $O -name "*kernel_trace.csv" -delete
