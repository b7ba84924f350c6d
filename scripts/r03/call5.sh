#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
GCNX_LIB=scripts/variants/libgcnx_tuning.so timeout -k 10 300 python3 scripts/wave_imbalance.py --workload powerlaw > $O/wave_imbalance.txt 2>&1; tail -12 $O/wave_imbalance.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_ecoli -- python3 bench.py --steps 200 --warmup 20 --cpu-seconds 0 --no-config3 > $O/trace_ecoli.log 2>&1; grep -c "SIGSEGV" $O/trace_ecoli.log
python3 bench.py --steps 200 --warmup 20 > $O/bench_ecoli_b.json 2> $O/bench_ecoli_b.err && python3 -c "
import json
for l in open('$O/bench_ecoli_b.json'):
    if l.startswith('{'):
        r=json.loads(l); print('ecoli', r['ms_per_step'], r['roofline']['frac'], 'config3', r['roofline_config3']['avg_launch_us'], r['roofline_config3']['frac'])
"
find $O -name "*kernel_trace.csv" -delete
