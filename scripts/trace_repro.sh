#!/bin/bash
# Product-free reproducer of the hipGraphLaunch SIGSEGV under rocprofv3 --kernel-trace (scripts/micro/graph_trace_repro.hip):
# each case once without the tracer and once with it; a SIGSEGV only kills the process.  Results -> gpurun_out/trace_repro/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/trace_repro; mkdir -p $O
B=scripts/micro/bin/graph_trace_repro
run() {  # name nodes launches variant sync_every
  name=$1; shift
  timeout -k 10 120 $B "$@" > $O/$name.plain.log 2>&1; rc0=$?
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -- $B "$@" > $O/$name.traced.log 2>&1; rc1=$?
  find $O/$name -name "*kernel_trace.csv" -delete 2>/dev/null
  echo "$name args=[$*] plain_rc=$rc0 traced_rc=$rc1 segv=$(grep -c 'SIGSEGV\|Segmentation' $O/$name.traced.log) ok=$(grep -c '^ok' $O/$name.traced.log)" | tee -a $O/summary.txt
}
: > $O/summary.txt
run shallow      150   20 1 0
run deep_linear  150  400 1 0
run deep_sync8   150  400 1 8
run deep_fork    150  400 3 0
run deep_bigarg  150  400 5 0
run deep_two     150  400 9 0
run tiny_deep      5 8000 1 0
echo done
