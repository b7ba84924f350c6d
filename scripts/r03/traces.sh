#!/bin/bash
# kernel traces of the four bench lines (end of round) + the scaling proxy
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_ecoli -- python3 bench.py --steps 200 --warmup 20 --cpu-seconds 0 --no-config3 > $O/trace_ecoli.log 2>&1; echo "ecoli rc=$?"
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_block1m -- python3 bench.py --workload block1m --steps 20 --warmup 3 --cpu-seconds 0 > $O/trace_block1m.log 2>&1; echo "block1m rc=$?"
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_powerlaw -- python3 bench.py --workload powerlaw --steps 20 --warmup 3 --cpu-seconds 0 > $O/trace_powerlaw.log 2>&1; echo "powerlaw rc=$?"
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_generalgnn -- python3 bench.py --model generalgnn --prec bf16x3 --steps 50 --warmup 5 --cpu-seconds 0 > $O/trace_generalgnn.log 2>&1; echo "generalgnn rc=$?"
find $O -name "*kernel_trace.csv" -delete
timeout -k 10 900 python3 scripts/scaling_proxy.py --out gpurun_out/scaling_proxy.json 2>&1 | tee gpurun_out/scaling_proxy.log | tail -5
