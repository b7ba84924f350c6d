#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_ecoli -- python3 -X faulthandler bench.py --steps 200 --warmup 20 --cpu-seconds 0 --no-config3 > $O/trace_ecoli.log 2>&1; echo "trace rc=$? coredump=$(grep -c 'GPU core' $O/trace_ecoli.log) segv=$(grep -c SIGSEGV $O/trace_ecoli.log)"
find $O -name "*kernel_trace.csv" -delete
grep -o '"generalgnn": {[^}]*}[^}]*}[^}]*}' $O/trace_ecoli.log | cut -c1-300
