#!/bin/bash
# Per-kernel stats of one rank's step of an 8-rank config-4 run (bench.py --emulate-rank), next to the one-GPU step
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/shard8; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 bench.py --workload block1m --emulate-rank 0 --of 8 --steps 100 --warmup 10 --cpu-seconds 0 > $O/t.log 2>&1
python3 scripts/kstats.py $O/t > $O/kstats_rank0of8.txt; tail -1 $O/t.log | head -c 600 >> $O/kstats_rank0of8.txt
python3 scripts/ktimeline.py $O/t > $O/timeline_rank0of8.txt 2>&1
find $O -name "*kernel_trace.csv" -delete
cat $O/kstats_rank0of8.txt
