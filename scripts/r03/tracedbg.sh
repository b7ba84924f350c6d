#!/bin/bash
# which part of the config-2 bench makes rocprofv3 --kernel-trace fall over (SIGSEGV inside hipGraphLaunch)?
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/tracedbg; mkdir -p $O
run() { name=$1; shift; timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -- python3 -X faulthandler bench.py "$@" > $O/$name.log 2>&1; echo "$name rc=$? segv=$(grep -c SIGSEGV $O/$name.log)"; }
run gnn_f32 --model generalgnn --prec f32 --steps 20 --warmup 3 --cpu-seconds 0
run with_gnn --steps 20 --warmup 3 --cpu-seconds 0 --no-config3
find $O -name "*kernel_trace.csv" -delete
grep -n "File \"\|Fatal Python\|Current thread" $O/with_gnn.log | head -20
grep -n "File \"\|Fatal Python\|Current thread" $O/gnn_f32.log | head -20
