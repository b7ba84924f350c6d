#!/bin/bash
# gpurun with a wait for a free slot: exit code 3 = "no box or slot free right now (nothing charged)" is retried every two
# minutes for up to GPU_WAIT_MIN minutes (default 40); every other outcome is returned as is.  Never retries a command that ran.
#   scripts/gpu.sh <timeout-seconds> '<command>'
T=$1; shift
for i in $(seq 1 $(( ${GPU_WAIT_MIN:-40} / 2 ))); do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 120
done
exit 3
